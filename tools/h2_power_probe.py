#!/usr/bin/env python3
"""Tuning aid (GPU box): the f16x2 GEMM on N(0,1) operands against the SAME launch on all-zero operands (identical instruction
stream and memory traffic, no switching activity in the multipliers): what the sustained clock under load costs."""
import os, sys, json, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from rag4dyg_amd import ops
dev = torch.device("cuda:0")
for M, K, N in ((63232, 512, 1536), (63232, 2048, 512), (8192, 8192, 8192)):
    out = {"M": M, "K": K, "N": N}
    for kind in ("normal", "zeros", "normal_again"):
        x = torch.randn(M, K, device=dev) if kind != "zeros" else torch.zeros(M, K, device=dev)
        w = torch.randn(K, N, device=dev) * 0.02 if kind != "zeros" else torch.zeros(K, N, device=dev)
        b = torch.zeros(N, device=dev)
        p = ops.split2_planes(w)
        for _ in range(5):
            ops.conv1d_h2(x, p, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 30 if M * K * N < 2e11 else 8
        e0.record()
        for _ in range(n):
            ops.conv1d_h2(x, p, b)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        out[kind + "_us"] = round(ms * 1e3, 1)
        out[kind + "_TF"] = round(2.0 * M * K * N / ms / 1e9, 1)
    print(json.dumps(out), flush=True)
