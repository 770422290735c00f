#!/bin/bash
# GPU box: rocprofv3 evidence for the Jaccard kernel: kernel trace + FETCH_SIZE / WRITE_SIZE (one counter per pass).
#   tools/profile_jaccard.sh <tag>   ->  gpurun_out/prof_jaccard_<tag>/{kernel_stats.csv,pmc.txt}
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_jaccard_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/bench_components.py jaccard > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
for grp in FETCH_SIZE WRITE_SIZE; do
  echo "pass $grp" >> $OUT/progress.log
  timeout -k 5 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$grp -- python3 $R/tools/bench_components.py jaccard > $OUT/pmc_$grp.log 2>&1 || echo "pass $grp FAILED" >> $OUT/progress.log
done
python3 - > $OUT/pmc.txt <<PY
import csv, glob, collections
print("# rocprofv3 --pmc <counter> -- python3 tools/bench_components.py jaccard ; per-dispatch values of jaccard_lds_kernel in launch order")
print("# (5 cases x (1 + reps) launches: hepth out x out, in x in, test x train, synthetic 20k out, 20k in); FETCH_SIZE / WRITE_SIZE in KiB;")
print("# hbm bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md)")
for grp in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = collections.defaultdict(list)
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % grp, recursive=True):
        for r in csv.DictReader(open(f)):
            if "jaccard" in r["Kernel_Name"]:
                vals[(r["Grid_Size"], r["Workgroup_Size"])].append(float(r["Counter_Value"]))
    for k, v in sorted(vals.items()):
        print(grp, "grid", k[0], "wg", k[1], "launches", len(v), "mean", round(sum(v) / len(v), 1))
PY
cat $OUT/pmc.txt; grep -i jaccard $OUT/kernel_stats.csv
