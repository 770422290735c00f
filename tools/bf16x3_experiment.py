#!/usr/bin/env python3
"""EXPERIMENT, not product (optional): what a 3-way bf16 split of both operands (6 products, fp32
accumulation) would buy over the exact-f32 MFMA GEMM on the bench's dominant launch shape, and what it costs in accuracy.
The bf16 products run on the vendor library through torch (plain library GEMMs), so the time is an upper bound on the idea's
cost with an unfused split and six separate launches; the error is the number that decides whether it is usable at all under
the 1e-4 tolerance.  The exact-f32 kernel stays the metric.      python tools/bf16x3_experiment.py
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rag4dyg_amd import ops                                   # noqa: E402

dev = torch.device("cuda:0")
M, K, N = 70912, 512, 2048                                    # c_fc of the UCI_13 bench step
g = torch.Generator().manual_seed(0)
A = (torch.randn(M, K, generator=g) * 1.3).to(dev)            # LayerNorm output scale
W = (torch.randn(K, N, generator=g) * 0.02).to(dev)           # Conv1D weight [in,out]
bias = torch.zeros(N, device=dev)


def split3(x):
    x0 = x.to(torch.bfloat16)
    r1 = x - x0.float()
    x1 = r1.to(torch.bfloat16)
    x2 = (r1 - x1.float()).to(torch.bfloat16)
    return x0, x1, x2


def mm32(a, b):
    try:
        return torch.mm(a, b, out_dtype=torch.float32)
    except TypeError:
        return None


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


out = {"experiment": "bf16x3 split GEMM vs exact-f32 MFMA GEMM", "shape": [M, K, N]}
ref_rows = torch.arange(0, M, M // 64)[:64]
ref = (A[ref_rows].double().cpu() @ W.double().cpu())
exact = ops.conv1d(A, W, bias)
out["f32_mfma_us"] = round(1e6 * timeit(lambda: ops.conv1d(A, W, bias)), 1)
err = lambda c: float(((c[ref_rows].double().cpu() - ref).abs().max() / ref.abs().max()))
out["f32_mfma_err_vs_f64"] = err(exact)
b0, b1, b2 = split3(W)
probe = mm32(A[:8].to(torch.bfloat16), b0)
if probe is None:
    out["bf16x3"] = "torch.mm(out_dtype=float32) not available in this torch build: products would round to bf16"
else:
    def emu():
        a0, a1, a2 = split3(A)                                 # the activations change every call; the weights are split once
        lo = mm32(a1, b1) + mm32(a0, b2) + mm32(a2, b0)
        mid = mm32(a0, b1) + mm32(a1, b0)
        return mm32(a0, b0) + (mid + lo)
    c = emu()
    out["bf16x3_err_vs_f64"] = err(c)
    out["bf16x3_us_vendor_unfused"] = round(1e6 * timeit(emu), 1)
    a0, a1, a2 = split3(A)
    out["bf16_one_product_us_vendor"] = round(1e6 * timeit(lambda: mm32(a0, b0)), 1)
    out["bf16x1_err_vs_f64"] = err(mm32(a0, b0))
    c2 = mm32(a0, b0) + (mm32(a0, b1) + mm32(a1, b0))
    out["bf16x2_3products_err_vs_f64"] = err(c2)
print(json.dumps(out))
