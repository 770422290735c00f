#!/bin/bash
# rocprofv3 kernel stats of the cached greedy decode step (tools/decode_step_bench.py: L6 H8 d768, 32 sequences).  Run through gpurun.
OUT=${1:-gpurun_out/decode_prof}
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o decode -- python3 tools/decode_step_bench.py > $OUT/wall.log 2> $OUT/err.log
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
cp "$f" $OUT/decode_kernel_stats.csv
cut -c1-150 $OUT/decode_kernel_stats.csv | head -14
tail -1 $OUT/wall.log
