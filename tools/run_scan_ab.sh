#!/bin/bash
# GPU box: scan + top-k variants at the north-star sizes (one process per variant: the tuning switches are read once).
#   tools/run_scan_ab.sh  ->  gpurun_out/scan_ab.jsonl
set -o pipefail
out=gpurun_out/scan_ab.jsonl
: > $out
run() { echo "# $*" >> $out; env "$@" python tools/bench_components.py scan >> $out 2>> gpurun_out/scan_ab.err || echo "FAILED $*" >> $out; }
run R4D_SCAN_CASES=12500x512,100000x512                              # default: 8-way split, occupancy-sized grid
run R4D_SCAN_CASES=12500x512,100000x512 R4D_SCAN_KW8=0                 # 4-way split (three workgroups per CU)
run R4D_SCAN_CASES=12500x512,100000x512 R4D_SCAN_KW8=0 R4D_SCAN_WGS_PER_CU=2
run R4D_SCAN_CASES=12500x512,100000x512 R4D_SCAN_WGS_PER_CU=1
run R4D_SCAN_CASES=12500x768,100000x768
cat $out
