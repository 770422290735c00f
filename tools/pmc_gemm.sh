#!/bin/bash
# Tuning aid (GPU box): PMC counters of the GEMM microbench, one counter group per pass (no tracing domains).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export R4D_GEMM_TILE=${1:-1} R4D_SHAPES=${2:-0}
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_BUSY_CU_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_t${R4D_GEMM_TILE}_s${R4D_SHAPES}/$tag -- python3 $R/tools/gemm_bench.py child > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
root="$R/gpurun_out/pmc_t${R4D_GEMM_TILE}_s${R4D_SHAPES}"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root+"/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "gemm_f32" in row["Kernel_Name"]:
            agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()):
        print(f"   {c:28s} mean {sum(vals)/len(vals):16.1f}  n={len(vals)}")
PY
