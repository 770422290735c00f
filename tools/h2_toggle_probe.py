#!/usr/bin/env python3
"""Tuning aid (GPU box): how much of the f16x2 GEMM's time follows the switching activity of its operands.  Same launch (c_attn shape
of the bench step), W's second-term plane (lo') (a) as it is, (b) with its low 3 mantissa bits cleared, (c) with its low 6 bits
cleared, (d) all zero; and the activations zero beside real weights.  Timing only: (b)-(d) are NOT the product arithmetic."""
import json, os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from rag4dyg_amd import ops
dev = torch.device("cuda:0")
M, K, N = 63232, 512, 1536
x = torch.randn(M, K, device=dev)
w = torch.randn(K, N, device=dev) * 0.02
b = torch.zeros(N, device=dev)
base = ops.split2_planes(w)


def timed(xx, planes, n=30):
    for _ in range(5):
        ops.conv1d_h2(xx, planes, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.conv1d_h2(xx, planes, b)
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)


out = {"as_is": timed(x, base)}
for name, mask in (("lo_low3_cleared", -8), ("lo_low6_cleared", -64), ("lo_zero", 0)):
    p = base.clone()
    p[:, :, 1, :] &= mask
    out[name] = timed(x, p)
p = base.clone(); p[:, :, 0, :] = 0
out["hi_zero"] = timed(x, p)
out["x_zero"] = timed(torch.zeros_like(x), base)
out["x_small_ints"] = timed(torch.randint(-2, 3, (M, K), device=dev).float(), base)      # lo' of the activations is zero, hi has 2 significant bits
out["as_is_again"] = timed(x, base)
print(json.dumps(out))
