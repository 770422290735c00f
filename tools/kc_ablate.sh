#!/bin/bash
# Tuning aid (run HERE, cross-compiles): builds ablated copies of the library (gemm_f32_kc.hip with -DKC_DBG=n; results
# are WRONG by construction) into tools/_bin/ so the GPU box can time what each part of the k-loop costs:
#   R4D_LIB_PATH=tools/_bin/librag4dyg_dbg<n>.so R4D_WT=1 R4D_SHAPES=14 python tools/gemm_bench.py 1
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_bin
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -DKC_DBG=$n -c rag4dyg_amd/csrc/gemm_f32_kc.hip -o tools/_bin/kc_dbg$n.o
  objs=$(ls rag4dyg_amd/_build/*.o | grep -v gemm_f32_kc.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/librag4dyg_dbg$n.so $objs tools/_bin/kc_dbg$n.o
done
