#!/usr/bin/env python3
"""Yardstick only (never used by the product): what the vendor fp32 GEMM (torch.matmul -> rocBLAS/hipBLASLt) reaches on
the encoder's shapes, next to r4d_conv1d_f32.  A ceiling claim needs a known-good reference on the same hardware."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from rag4dyg_amd import ops
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
SHAPES = [(8864, 512, 1536), (35456, 512, 1536), (35456, 512, 2048), (35456, 2048, 512), (35456, 512, 512), (16384, 1024, 4096)]
if os.environ.get("R4D_YARD_SHAPES"):
    SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ["R4D_YARD_SHAPES"].split(",")]
for M, K, N in SHAPES:
    x = torch.randn(M, K, device=dev); w = torch.randn(K, N, device=dev) * 0.02; b = torch.randn(N, device=dev)
    res = []
    for fn in (lambda: torch.addmm(b, x, w), lambda: ops.conv1d(x, w, b)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20)
    f = 2.0 * M * K * N / 1e9
    print(f"M={M:6d} K={K:5d} N={N:5d}  torch.addmm {res[0]*1e3:8.1f} us {f/res[0]:7.1f} TF   r4d {res[1]*1e3:8.1f} us {f/res[1]:7.1f} TF", flush=True)
