#!/bin/bash
# One-GPU rehearsal of the multi-rank bench: N processes on cuda:0, collectives over gloo (R4D_BENCH_BACKEND=gloo), the pipelined
# sharded top-k forced on.  Not a performance number (the ranks share one GPU): what it records is that the sharded, merged
# top-k of a timed step equals the one-GPU recomputation at every N, with the shard sizes an N-GPU run would have.
#   tools/run_gloo_rehearsal.sh "2 4 5"   ->  gpurun_out/rehearsal_gloo.jsonl
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 R4D_BENCH_BACKEND=gloo R4D_BENCH_PIPELINE=force
out=gpurun_out/rehearsal_gloo.jsonl; : > $out
for N in ${1:-2 4 5}; do
  timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600+N)) \
      bench.py --gpus $N --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/rehearsal_$N.json 2> gpurun_out/rehearsal_$N.err
  python - >> $out <<PY
import json
try:
    d = [json.loads(l) for l in open("gpurun_out/rehearsal_$N.json") if l.startswith("{")][-1]
    v = d["extras"]["verify"]; sq = d["extras"]["scan_q32"]
    print(json.dumps({"ranks": $N, "backend": "gloo, all ranks on one GPU (rehearsal, not a scaling number)", "shard_rows": sq["pool_rows"],
                      "sharded_topk_equals_one_gpu": v.get("sharded_topk_equals_one_gpu"), "world_size": v.get("world_size"),
                      "error": v.get("error"), "rank_spread": v.get("rank_spread"), "scan_kernel_us_on_shard": sq["scan_kernel_us"],
                      "source_sha": d["extras"]["source_sha"]}))
except Exception as e:
    print(json.dumps({"ranks": $N, "error": str(e)}))
PY
  sleep 3
done
cat $out
