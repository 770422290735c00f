#!/bin/bash
# GPU box: scan + top-k variants at the north-star sizes (one process per variant: the tuning switches are read once).
#   tools/run_scan_ab.sh  ->  gpurun_out/scan_ab.jsonl
set -o pipefail
out=gpurun_out/scan_ab.jsonl
: > $out
run() { echo "# $*" >> $out; env "$@" python tools/bench_components.py scan 2>> gpurun_out/scan_ab.err | grep '"component": "scan"' >> $out || echo "FAILED $*" >> $out; }
run R4D_SCAN_CASES=12500x512,100000x512                              # default: rows dealt evenly over the occupancy-sized grid
run R4D_SCAN_CASES=12500x512,100000x512 R4D_SCAN_WGS_PER_CU=1          # one workgroup per CU: 49 rows each, two tiles in flight
run R4D_SCAN_CASES=12500x512,100000x512 R4D_SCAN_WGS_PER_CU=1 R4D_SCAN_TWO=0
run R4D_SCAN_CASES=12500x512,100000x512 R4D_SCAN_WGS_PER_CU=3
run R4D_SCAN_CASES=12500x768,100000x768
run R4D_SCAN_CASES=12500x768,100000x768 R4D_SCAN_WGS_PER_CU=1
cat $out
