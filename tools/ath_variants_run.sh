#!/bin/bash
# Tuning aid (GPU box): time compile-time variants of attention_h2.hip built into tools/_bin/librag4dyg_<name>.so
#   tools/ath_variants_run.sh "1 3" k8v3o3 k16v6o2 ...
cd "$(dirname "$0")/.."
export ATT_H2_ONLY=1 R4D_ALLOW_ABLATED_LIB=1 R4D_SHAPES=$(echo $1 | tr ' ' ','); shift
for n in "$@"; do
  export R4D_LIB_PATH=$PWD/tools/_bin/librag4dyg_$n.so
  timeout -k 10 120 python3 tools/attn_bench.py 2>/dev/null || exit 1
done
