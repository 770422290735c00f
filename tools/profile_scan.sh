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
export R4D_SCAN_CASES=${R4D_SCAN_CASES:-12512x512,100000x512,100000x768}
# one pool size per process, so that the per-kernel averages and the per-launch counter means are not a mix of sizes
for size in ${R4D_SCAN_CASES//,/ }; do
  echo "trace $size" >> $OUT/progress.log
  R4D_SCAN_CASES=$size R4D_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$size -- python3 $R/tools/bench_components.py scan > $OUT/trace_$size.log 2>&1
  find $OUT/trace_$size -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$size.csv
  for grp in FETCH_SIZE WRITE_SIZE; do
    echo "pass $size $grp" >> $OUT/progress.log
    R4D_SCAN_CASES=$size R4D_NO_GRAPH=1 timeout -k 5 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_${size}_$grp -- python3 $R/tools/bench_components.py scan > $OUT/pmc_${size}_$grp.log 2>&1 || echo "pass $size $grp FAILED" >> $OUT/progress.log
  done
done
python3 - > $OUT/pmc.txt <<PY
import csv, glob, collections
import os
print("# per-launch means, KiB as reported; hbm bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction)")
for d in sorted(glob.glob("$OUT/pmc_*_*")):
    if not os.path.isdir(d):
        continue
    size, grp = os.path.basename(d)[4:].split("_", 1)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    for k, (v, n) in sorted(agg.items()):
        if "scan" in k or "topk" in k:
            print(size, grp, k, "launches", n, "mean", round(v / n, 1))
PY
# profiles/pmc_scan.json: what bench.py's extras.scan_q32 reports as `traffic` (stamped with the sources it was measured on)
python3 - > $OUT/pmc_scan.json <<PY
import json, re, sys
sys.path.insert(0, "$R")
from bench import source_sha
out = {"_workload": {"source_sha": source_sha(), "queries": 32,
                     "command": "tools/profile_scan.sh: R4D_SCAN_CASES=<size> rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 tools/bench_components.py scan (one counter and one size per pass)"}}
vals = {}
for line in open("$OUT/pmc.txt"):
    m = re.match(r"(\d+x\d+) (FETCH_SIZE|WRITE_SIZE) (.+?) launches \d+ mean ([0-9.]+)", line)
    if m and "pool_scan" in m.group(3):
        vals.setdefault(m.group(1), {})[m.group(2)] = float(m.group(4))
for size, v in vals.items():
    if len(v) == 2:
        out[size] = {"fetch_kib_raw": v["FETCH_SIZE"], "write_kib": v["WRITE_SIZE"],
                     "hbm_bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024}
print(json.dumps(out, indent=1))
PY
cat $OUT/pmc.txt; for f in $OUT/kernel_stats_*.csv; do echo "== $f"; grep -i "scan\|topk\|Name" $f; done
