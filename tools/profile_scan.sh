#!/bin/bash
# GPU box: rocprofv3 evidence for the pool scan + top-k (the north-star HBM-bound kernel).
#   pass 1: --kernel-trace --stats   -> per-kernel durations
#   pass 2/3: --pmc FETCH_SIZE / WRITE_SIZE (separate passes, no tracing domains) -> HBM traffic
# Usage: tools/profile_scan.sh <tag>     outputs under gpurun_out/prof_scan_<tag>/
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_scan_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export R4D_SCAN_CASES=${R4D_SCAN_CASES:-12500x512,100000x512}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/bench_components.py scan topk > $OUT/trace.log 2>&1
for grp in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$grp -- python3 $R/tools/bench_components.py scan > $OUT/pmc_$grp.log 2>&1
done
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
python3 - <<PY
import csv, glob, collections
for grp in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % grp, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    for k, (v, n) in sorted(agg.items()):
        if "scan" in k or "topk" in k:
            print(grp, k, "launches", n, "avg per launch", v / n)
PY
head -30 $OUT/kernel_stats.csv
