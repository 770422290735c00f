#!/bin/bash
# Tuning aid (GPU box): time the ablated builds of gemm_s3.hip (tools/kc_ablate.sh gemm_s3.hip S3_DBG n...) on a few shapes.
#   tools/s3_ablate_run.sh "0 2 8" 0 1 2 4 8 16      (shape indices of tools/s3_bench.py, then the S3_DBG values; 0 = product)
cd "$(dirname "$0")/.."
export S3_TIME_ONLY=1 R4D_ALLOW_ABLATED_LIB=1 R4D_SHAPES=$(echo $1 | tr ' ' ','); shift
export R4D_GEMM_S3_TILE=${S3_TILE:-0}
for n in "$@"; do
  if [ "$n" = 0 ]; then unset R4D_LIB_PATH; else export R4D_LIB_PATH=$PWD/tools/_bin/librag4dyg_dbg$n.so; fi
  timeout -k 10 120 python3 tools/s3_bench.py child 2>/dev/null || exit 1
done
