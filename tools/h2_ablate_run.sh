#!/bin/bash
# Tuning aid (GPU box): time the ablated builds of gemm_h2.hip (tools/kc_ablate.sh gemm_h2.hip H2_DBG n...) on a few shapes.
#   [KIND=h2p] tools/h2_ablate_run.sh "0 2 8" 0 1 2 4 8 32 64      (shape indices of tools/s3_bench.py, then the H2_DBG values; 0 = product)
cd "$(dirname "$0")/.."
export S3_TIME_ONLY=1 S3_BENCH_KIND=${KIND:-h2} R4D_ALLOW_ABLATED_LIB=1 R4D_SHAPES=$(echo $1 | tr ' ' ','); shift
for n in "$@"; do
  if [ "$n" = 0 ]; then unset R4D_LIB_PATH; else export R4D_LIB_PATH=$PWD/tools/_bin/librag4dyg_dbg$n.so; fi
  timeout -k 10 120 python3 tools/s3_bench.py child 2>/dev/null || exit 1
done
