#!/bin/bash
# Tuning aid (GPU box): PMC counters of the bf16x3 GEMM microbench (tools/s3_bench.py, timing-only mode), one counter
# group per pass (no tracing domains).   tools/pmc_s3.sh <tile> "<shape indices>"      (S3_BENCH_KIND=h2: the f16x2 GEMM)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export R4D_GEMM_S3_TILE=${1:-0} R4D_SHAPES=$(echo ${2:-8} | tr ' ' ',') S3_TIME_ONLY=1
OUT=$R/gpurun_out/pmc_${S3_BENCH_KIND:-s3}_t${R4D_GEMM_S3_TILE}
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/$tag -- python3 $R/tools/s3_bench.py child > $OUT.$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "gemm_s3" in row["Kernel_Name"] or "gemm_h2" in row["Kernel_Name"]:
            agg[row["Kernel_Name"][:70] + " grid " + row.get("Grid_Size", "?")][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()):
        print(f"   {c:28s} mean {sum(vals)/len(vals):16.1f}  n={len(vals)}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
        m=sum(v["SQ_VALU_MFMA_BUSY_CYCLES"])/len(v["SQ_VALU_MFMA_BUSY_CYCLES"]); g=sum(v["GRBM_GUI_ACTIVE"])/len(v["GRBM_GUI_ACTIVE"])
        print(f"   mfma_pipe_util {m/1024/(g/8):.3f}   cycles/XCD {g/8:.0f}")
PY
