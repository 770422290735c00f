#!/bin/bash
# GPU box: everything that ends up under profiles/ for a round, in FOUR calls (a gpurun call is at most 20 minutes):
#   tools/run_round_evidence.sh r04 pmc        rocprofv3 traces + PMC passes of bench.py in its three arithmetics
#   tools/run_round_evidence.sh r04 pmc2       ... on the wikiv2 shape, and of the scan
#   tools/run_round_evidence.sh r04 lines      installs those PMC summaries, then the bench lines (so that they carry `traffic`), acceptance
#                                              table, GEMM shapes, end-to-end CLI timings
#   tools/run_round_evidence.sh r04 rest       components, Jaccard, training step, decode step
# Each step appends to gpurun_out/evidence_<tag>.log so that a long run never looks hung.  Everything to commit is collected under
# gpurun_out/evidence_<tag>/ : copy that directory's files into profiles/.
TAG=${1:-r04}
PART=${2:-pmc}
R=${GRAFT_REPO_ROOT:-/root/repo}
LOG=$R/gpurun_out/evidence_$TAG.log
EV=$R/gpurun_out/evidence_$TAG
mkdir -p $EV
cd $R
if [ "$PART" = pmc ]; then
  echo "== profile bench UCI_13 (f16x2 GEMMs: the headline)" >> $LOG
  R4D_PROFILE_GEMM=f16x2 bash tools/profile_bench.sh $TAG >> $LOG 2>&1
  echo "== profile bench UCI_13 (bf16x3 GEMMs)" >> $LOG
  R4D_PROFILE_GEMM=split3 bash tools/profile_bench.sh ${TAG}s >> $LOG 2>&1
  echo "== profile bench UCI_13 (exact-f32 MFMA GEMMs)" >> $LOG
  R4D_PROFILE_GEMM=f32 bash tools/profile_bench.sh ${TAG}f >> $LOG 2>&1
  cp gpurun_out/profiles_$TAG/* gpurun_out/profiles_${TAG}s/* gpurun_out/profiles_${TAG}f/* $EV/
  echo "== pmc part done" >> $LOG
elif [ "$PART" = pmc2 ]; then
  echo "== profile bench wikiv2 (f16x2)" >> $LOG
  R4D_PROFILE_GEMM=f16x2 R4D_PROFILE_SHAPE=wikiv2 bash tools/profile_bench.sh ${TAG}w >> $LOG 2>&1
  echo "== profile scan" >> $LOG
  bash tools/profile_scan.sh $TAG >> $LOG 2>&1
  cp gpurun_out/profiles_${TAG}w/* $EV/
  cp gpurun_out/prof_scan_$TAG/pmc_scan.json $EV/pmc_scan.json
  cp gpurun_out/prof_scan_$TAG/pmc.txt $EV/${TAG}_scan_pmc.txt
  for f in gpurun_out/prof_scan_$TAG/kernel_stats_*.csv; do cp $f $EV/${TAG}_scan_$(basename $f); done
  echo "== pmc2 part done" >> $LOG
elif [ "$PART" = lines ]; then
  # the PMC summaries of the pmc part (copied into profiles/ by the builder, or still under gpurun_out/ in the same snapshot)
  for pair in "${TAG}_pmc_traffic.json:pmc_traffic_f16x2.json" "${TAG}s_pmc_traffic.json:pmc_traffic.json" "${TAG}f_pmc_traffic.json:pmc_traffic_f32.json" \
              "${TAG}w_pmc_traffic.json:pmc_traffic_wikiv2_f16x2.json" "pmc_scan.json:pmc_scan.json"; do
    src=${pair%%:*}; dst=${pair##*:}
    [ -f $EV/$src ] && cp $EV/$src profiles/$dst
    [ -f profiles/$src ] && [ "$src" != "$dst" ] && cp profiles/$src profiles/$dst
  done
  echo "== bench UCI_13" >> $LOG
  python bench.py > $EV/${TAG}_bench_line.json 2>> $LOG
  echo "== bench wikiv2" >> $LOG
  python bench.py --shape wikiv2 > $EV/${TAG}_bench_line_wikiv2.json 2>> $LOG
  echo "== acceptance table + GEMM shapes + end-to-end CLI timings" >> $LOG
  python tools/s3_acceptance.py > $EV/${TAG}_h2_acceptance.md 2>> $LOG
  python tools/s3_bench.py auto 2>> $LOG | grep "^{" > $EV/${TAG}_gemm_shapes.jsonl
  python tools/attn_acceptance.py > $EV/${TAG}_attention_h2_acceptance.md 2>> $LOG
  python tools/h2_power_probe.py 2>> $LOG | grep "^{" > $EV/${TAG}_h2_power_probe.jsonl
  python tools/annotation_e2e.py annotation retriever 2>> $LOG | grep "^{" > $EV/${TAG}_cli_end_to_end.jsonl
  cp profiles/pmc_traffic_f16x2.json profiles/pmc_traffic.json profiles/pmc_traffic_f32.json profiles/pmc_traffic_wikiv2_f16x2.json profiles/pmc_scan.json $EV/ 2>/dev/null
  echo "== lines part done" >> $LOG
else
  echo "== components" >> $LOG
  python tools/bench_components.py > $EV/${TAG}_components.jsonl 2>> $LOG
  echo "== profile jaccard" >> $LOG
  bash tools/profile_jaccard.sh $TAG >> $LOG 2>&1
  cp gpurun_out/prof_jaccard_$TAG/pmc.txt $EV/${TAG}_jaccard_pmc.txt
  cp gpurun_out/prof_jaccard_$TAG/kernel_stats.csv $EV/${TAG}_jaccard_kernel_stats.csv
  echo "== profile training step" >> $LOG
  bash tools/profile_training.sh gpurun_out/train_prof >> $LOG 2>&1
  cp gpurun_out/train_prof/train_kernel_stats.csv $EV/${TAG}_training_step_kernel_stats.csv
  tail -1 gpurun_out/train_prof/line.json > $EV/${TAG}_training_step.jsonl
  echo "== profile decode step" >> $LOG
  bash tools/profile_decode.sh gpurun_out/decode_prof >> $LOG 2>&1
  cp gpurun_out/decode_prof/decode_kernel_stats.csv $EV/${TAG}_decode_step_kernel_stats.csv
  echo "== rest part done" >> $LOG
fi
ls $EV >> $LOG
