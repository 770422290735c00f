set -o pipefail
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "scan or score or topk" > gpurun_out/scan_s3_tests.log 2>&1; tail -3 gpurun_out/scan_s3_tests.log
out=gpurun_out/scan_s3_ab.jsonl; : > $out
run() { echo "# $*" >> $out; env "$@" python tools/bench_components.py scan 2>> gpurun_out/scan_s3_ab.err | grep '"component": "scan"' >> $out || echo "FAILED $*" >> $out; }
run R4D_SCAN_CASES=12500x512,100000x512,12500x768,100000x768
run R4D_SCAN_CASES=12500x512,100000x512,12500x768,100000x768 R4D_GEMM_SPLIT3=0
run R4D_SCAN_CASES=12500x512,100000x512 R4D_SCAN_TWO=0
run R4D_SCAN_CASES=12500x512,100000x512 R4D_SCAN_KW8=0
run R4D_SCAN_CASES=12500x512,100000x512 R4D_SCAN_WGS_PER_CU=2
python - <<'PY'
import json
for l in open('gpurun_out/scan_s3_ab.jsonl'):
    if l.startswith('#'): print(l.strip()); continue
    try: r=json.loads(l)
    except Exception: print(l.strip()); continue
    print(r['N'],r['d'],r['operands'],'scan',r['kernel_us'],'frac',r['roofline']['frac'],'topk',r['topk_us'],'gpu_wall',r['gpu_wall_us'])
PY
