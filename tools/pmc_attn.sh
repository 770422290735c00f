#!/bin/bash
# Tuning aid (GPU box): PMC counters of the attention microbench.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_attn/$tag -- python3 $R/tools/attn_bench.py > /dev/null 2>&1   # R4D_SHAPES selects the shape
done
python3 - <<PY
import csv, glob, collections
root="$R/gpurun_out/pmc_attn"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root+"/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "attn_colsplit" in row["Kernel_Name"]:
            agg[row["Kernel_Name"][:50]+" grid "+row["Grid_Size"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()):
        print(f"   {c:28s} mean {sum(vals)/len(vals):16.1f}  n={len(vals)}")
PY
