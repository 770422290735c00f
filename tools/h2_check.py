#!/usr/bin/env python3
"""GPU box: the f16x2 GEMM (r4d_conv1d_h2_f32) against float64 on shapes that exercise interior and edge tiles of both tile
shapes, every epilogue, and the operand range (tiny activations, large activations, an outlier channel).  Prints max-norm and
element-wise errors of f16x2, bf16x3 and the exact-f32 kernel side by side.    python tools/h2_check.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rag4dyg_amd import ops                                   # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
bad = 0
for M, K, N in ((45, 96, 80), (128, 32, 256), (129, 64, 257), (1000, 512, 1536), (4096, 512, 512), (4096, 512, 1536), (700, 2048, 512), (333, 768, 2304)):
    for epi in ("none", "gelu", "residual"):
        for kind, scale in (("N(0,1)", 1.0), ("tiny 1e-3", 1e-3), ("large 3e4", 3e4), ("outlier", 1.0)):
            x = torch.randn(M, K, device=dev, generator=g) * scale
            if kind == "outlier":
                x[:, 3] *= 500
            w = torch.randn(K, N, device=dev, generator=g) * 0.02
            b = torch.randn(N, device=dev, generator=g) * (0.1 * scale)
            r = torch.randn(M, N, device=dev, generator=g) * scale if epi == "residual" else None
            ref = x.double() @ w.double() + b.double()
            if epi == "gelu":
                ref = 0.5 * ref * (1.0 + torch.tanh(0.7978845608028654 * (ref + 0.044715 * ref ** 3)))
            elif epi == "residual":
                ref = ref + r.double()
            pl2 = ops.split2_planes(w)                       # run to run: the same bits (a race would show here)
            y0 = ops.conv1d_h2(x, pl2, b, epi, r)
            if not all(torch.equal(y0, ops.conv1d_h2(x, pl2, b, epi, r)) for _ in range(3)):
                bad += 1
                print(f"   h2 NOT REPRODUCIBLE at {M}x{K}x{N} {epi} {kind}", flush=True)
            res = {}
            for name, fn in (("h2", lambda: ops.conv1d_h2(x, ops.split2_planes(w), b, epi, r)),
                             ("s3", lambda: ops.conv1d_s3(x, ops.split3_planes(w), b, epi, r)),
                             ("f32", lambda: ops.conv1d(x, w, b, epi, r, w.t().contiguous()))):
                y = fn().double()
                d = (y - ref).abs()
                res[name] = (float(d.max() / ref.abs().max()), float((d / (1e-4 * ref.abs() + 1e-5 * ref.abs().max())).max()))
            flag = "" if res["h2"][0] <= 1.5 * max(res["f32"][0], res["s3"][0]) + 1e-9 and res["h2"][1] < 1 else "   <-- CHECK"
            bad += bool(flag)
            print(f"{M:5d}x{K:4d}x{N:4d} {epi:8s} {kind:10s} h2 {res['h2'][0]:.2e}/{res['h2'][1]:.4f}  s3 {res['s3'][0]:.2e}/{res['s3'][1]:.4f}  "
                  f"f32 {res['f32'][0]:.2e}/{res['f32'][1]:.4f}{flag}", flush=True)
print(json.dumps({"cases_flagged": bad}))
