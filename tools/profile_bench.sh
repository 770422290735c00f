#!/bin/bash
# GPU box: rocprofv3 evidence for bench.py.  Pass 1: --kernel-trace --stats (per-kernel durations).
# Passes 2-4: PMC counters, one group per pass, no tracing domains (HBM traffic, L2 hit rate, MFMA busy).
# Usage: tools/profile_bench.sh <round-tag>          outputs under gpurun_out/prof_<tag>/ + a summary
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
export R4D_PROFILE_SHAPE=${R4D_PROFILE_SHAPE:-UCI_13}
# --headline-only: warm-up + timed steps and NOTHING else (no roofline re-run, no exact-f32 second run, no length-bucketed
# run), so that the per-kernel averages of this trace are the headline launches' (the r02 trace mixed them)
export R4D_PROFILE_GEMM=${R4D_PROFILE_GEMM:-f16x2}
GEMM=$R4D_PROFILE_GEMM
ARGS="--shape $R4D_PROFILE_SHAPE --gemm $GEMM --steps 32 --warmup 16 --random-pool --headline-only"
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# an ordinary run first (HIP-event roofline of the same steps): its flop_per_launch goes beside the rocprofv3 average
python3 $R/bench.py --shape $R4D_PROFILE_SHAPE --gemm $GEMM --steps 32 --warmup 16 --random-pool --no-cpu-baseline --no-bucketed --no-exact-f32 > $OUT.line.json 2>> $OUT.bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS >> $OUT.bench.log 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  echo "pmc pass $tag" >> $OUT/progress.log
  timeout -k 5 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$tag -- python3 $R/bench.py $ARGS > /dev/null 2>&1 || echo "pmc pass $tag FAILED" >> $OUT/progress.log
done
python3 $R/tools/profile_summary.py $OUT $TAG
