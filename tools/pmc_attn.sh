#!/bin/bash
# Tuning aid (GPU box): PMC counters of the f16x2 attention kernel (tools/attn_bench.py, ATT_H2_ONLY), one counter group per
# pass (no tracing domains).   tools/pmc_attn.sh "<shape indices>"
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export R4D_SHAPES=$(echo ${1:-1} | tr ' ' ',') ATT_H2_ONLY=1
OUT=$R/gpurun_out/pmc_attn
mkdir -p $OUT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD" \
           "TA_BUSY_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout -k 5 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/$tag -- python3 $R/tools/attn_bench.py > $OUT.$tag.log 2>&1 || echo "pass $tag failed"
done
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "attn_h2" in row["Kernel_Name"] or "attn_colsplit" in row["Kernel_Name"]:
            agg[row["Kernel_Name"][:60] + " grid " + row.get("Grid_Size", "?")][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()):
        print(f"   {c:44s} mean {sum(vals)/len(vals):16.1f}  n={len(vals)}")
PY
