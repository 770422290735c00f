#!/bin/bash
# GPU box: cache / HBM counters for the pool scan at ONE size (default 100000x512), ONE counter per pass.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_scan_${1:-x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export R4D_SCAN_CASES=${R4D_SCAN_CASES:-100000x512} R4D_NO_GRAPH=1
for grp in ${PMC_LIST:-FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES}; do
  echo "pass $grp" >> $OUT/progress.log
  timeout -k 5 90 rocprofv3 --pmc $grp --output-format csv -d $OUT/$grp -- python3 $R/tools/bench_components.py scan > $OUT/$grp.log 2>&1 || echo "pass $grp FAILED" >> $OUT/progress.log
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in sorted(agg.items()):
    if "scan" in k or "topk" in k:
        print(k, {c: round(v / n, 1) for c, (v, n) in sorted(d.items())})
PY
