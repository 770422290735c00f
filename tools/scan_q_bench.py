"""Scan + top-k at the operating points of the bench: Q = 32 / 256 x 100k rows (one GPU), Q = 2048 x 12,512 (what one
rank of an 8-GPU weak-scaling run does per step), d = 512 and 768.  HIP-event timed per call; R4D_SCAN_TILED=0 selects the 32-query
blocks for every Q (the round-4 form).  Prints one JSON line per case with the fp32-equivalent TFLOP/s against the bf16x3 ceiling
(2,500 / 6) and the GB/s of ONE read of the shard."""
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rag4dyg_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for d, cases in ((512, ((32, 100000), (256, 100000), (2048, 12512), (64, 12512), (128, 100000))), (768, ((256, 100000), (2048, 12512)))):
    for Q, N in cases:
        q = ops.normalize_rows(torch.randn(Q, d, generator=g).to(dev))
        p = ops.normalize_rows(torch.randn(N, d, generator=g).to(dev))
        for _ in range(3):
            ops.score_topk(q, p, 10)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            ops.score_topk(q, p, 10)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(json.dumps({"d": d, "Q": Q, "N": N, "tiled": os.environ.get("R4D_SCAN_TILED", "1"), "scan_plus_topk_us": round(us, 1),
                          "TFLOPs_fp32_equiv": round(2.0 * Q * N * d / us / 1e6, 1), "frac_of_bf16x3_ceiling": round(2.0 * Q * N * d / us / 1e6 / 416.7, 3),
                          "one_read_GBps": round(4.0 * N * d / us / 1e3, 1)}), flush=True)
