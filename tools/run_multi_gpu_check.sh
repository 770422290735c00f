#!/bin/bash
# First thing to run on a multi-GPU lease: bench.py over RCCL at N ranks (default 8; the launcher starts BEFORE any GPU
# call), then one line saying whether (1) the group really had N ranks on N devices, (2) the sharded, merged top-k of a
# timed step equals a one-GPU recomputation over the whole 100k pool bit for bit, (3) how unevenly the ranks were loaded.
# Then the STRONG-scaling form (the same 64 batches = 2,048 queries per step split over the ranks), and at N = 1 the
# one-GPU strong baseline, so that one lease yields: the weak number the north star quotes, a strong-scaling efficiency, the
# RCCL version and the two all-gathers' own times (extras.verify.collectives).
#   tools/run_multi_gpu_check.sh [N] [steps]        -> gpurun_out/multi_gpu_check_N.json (+ .err), multi_gpu_strong_{1,N}.json
N=${1:-8}
STEPS=${2:-20}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29555 \
    $R/bench.py --gpus $N --steps $STEPS --warmup 5 > $R/gpurun_out/multi_gpu_check_$N.json 2> $R/gpurun_out/multi_gpu_check_$N.err
rc=$?
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29556 \
    $R/bench.py --gpus $N --steps $STEPS --warmup 3 --scaling strong --no-verify > $R/gpurun_out/multi_gpu_strong_$N.json 2>> $R/gpurun_out/multi_gpu_check_$N.err
python $R/bench.py --gpus 1 --steps 5 --warmup 2 --scaling strong --no-cpu-baseline --no-bucketed --no-exact-f32 --no-roofline \
    > $R/gpurun_out/multi_gpu_strong_1.json 2>> $R/gpurun_out/multi_gpu_check_$N.err
python - <<PY
import json, sys
try:
    d = [json.loads(l) for l in open("$R/gpurun_out/multi_gpu_check_$N.json") if l.startswith("{")][-1]
except Exception as e:
    print("multi-GPU check: no bench line (rc=$rc):", e); sys.exit(1)
v = d["extras"]["verify"] or {}
print(f"multi-GPU check: n_gpus={d['n_gpus']} value={d['value']} {d['unit']}  collectives={d['config']['collectives']}")
print(f"  world_size={v.get('world_size')} devices={v.get('devices')} sharded_topk_equals_one_gpu={v.get('sharded_topk_equals_one_gpu')} "
      f"rank_spread={v.get('rank_spread')} per_rank_ms_per_step={v.get('per_rank_ms_per_step')} error={v.get('error')}")
print(f"  collectives: {v.get('collectives')}")
try:
    sN = [json.loads(l) for l in open("$R/gpurun_out/multi_gpu_strong_$N.json") if l.startswith("{")][-1]
    s1 = [json.loads(l) for l in open("$R/gpurun_out/multi_gpu_strong_1.json") if l.startswith("{")][-1]
    print(f"  strong scaling (2,048 queries per step): N=1 {s1['value']} -> N=$N {sN['value']} {sN['unit']}: "
          f"speed-up {sN['value'] / s1['value']:.2f}x, efficiency {sN['value'] / s1['value'] / $N:.2f}")
except Exception as e:
    print("  strong-scaling lines missing:", e)
sys.exit(0 if v.get("sharded_topk_equals_one_gpu") and len(v.get("devices", [])) == $N else 2)
PY
