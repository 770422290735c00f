#!/bin/bash
# Tuning aid (run HERE, cross-compiles): builds ablated copies of the library -- ONE source file recompiled with a debug
# macro that removes part of a kernel (results are WRONG by construction) -- into tools/_bin/, so the GPU box can time
# what each part costs.   usage: tools/kc_ablate.sh [file.hip MACRO] n...      (default: gemm_f32_kc.hip KC_DBG)
#   R4D_LIB_PATH=$PWD/tools/_bin/librag4dyg_dbg<n>.so R4D_WT=1 R4D_SHAPES=14 python tools/gemm_bench.py 1
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_bin
FILE=gemm_f32_kc.hip; MACRO=KC_DBG
if [[ "$1" == *.hip ]]; then FILE=$1; MACRO=$2; shift 2; fi
BASE=${FILE%.hip}
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -D$MACRO=$n -c rag4dyg_amd/csrc/$FILE -o tools/_bin/${BASE}_dbg$n.o
  objs=$(ls rag4dyg_amd/_build/*.o | grep -v "/$BASE.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/librag4dyg_dbg$n.so $objs tools/_bin/${BASE}_dbg$n.o
done
