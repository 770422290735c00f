#!/bin/bash
# rocprofv3 kernel stats of the retriever training step (tools/bench_components.py training).  Run through gpurun.
set -e
OUT=${1:-gpurun_out/train_prof}
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o train -- python3 tools/bench_components.py training > $OUT/line.json 2> $OUT/err.log
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
cp "$f" $OUT/train_kernel_stats.csv
head -30 $OUT/train_kernel_stats.csv | cut -c1-200
cat $OUT/line.json | tail -1
