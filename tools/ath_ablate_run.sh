#!/bin/bash
# Tuning aid (GPU box): time the ablated builds of attention_h2.hip (tools/kc_ablate.sh attention_h2.hip ATH_DBG n...).
#   tools/ath_ablate_run.sh "1 3" 0 1 2 4 8 16      (shape indices of tools/attn_bench.py, then the ATH_DBG values; 0 = product)
cd "$(dirname "$0")/.."
export ATT_H2_ONLY=1 R4D_ALLOW_ABLATED_LIB=1 R4D_SHAPES=$(echo $1 | tr ' ' ','); shift
for n in "$@"; do
  if [ "$n" = 0 ]; then unset R4D_LIB_PATH; else export R4D_LIB_PATH=$PWD/tools/_bin/librag4dyg_dbg$n.so; fi
  timeout -k 10 120 python3 tools/attn_bench.py 2>/dev/null || exit 1
done
