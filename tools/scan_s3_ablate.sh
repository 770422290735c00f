#!/bin/bash
# GPU box: where the shard-size scan's time goes -- phase timeline (SCAN_DBG=8 build) and ablated builds (tools/kc_ablate.sh score.hip SCAN_DBG 1 2 8 16)
set -o pipefail
export R4D_ALLOW_ABLATED_LIB=1
out=gpurun_out/scan_s3_ablate.txt; : > $out
for two in 1 0; do
  echo "## timeline R4D_SCAN_TWO=$two" >> $out
  R4D_SCAN_TWO=$two R4D_LIB_PATH=$PWD/tools/_bin/librag4dyg_dbg8.so python tools/scan_timeline.py 12500 512 >> $out 2>&1
done
for n in 0 1 16 2; do
  lib=$PWD/tools/_bin/librag4dyg_dbg$n.so; [ $n = 0 ] && lib=$PWD/rag4dyg_amd/librag4dyg_hip.so
  echo "## SCAN_DBG=$n" >> $out
  R4D_LIB_PATH=$lib R4D_NO_GRAPH=1 R4D_SCAN_CASES=12500x512,100000x512 python tools/bench_components.py scan 2>/dev/null | grep '"component": "scan"' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['N'], r['d'], r['operands'], 'scan_us', r['kernel_us'], 'frac', r['roofline']['frac'])" >> $out
done
cat $out
