#!/usr/bin/env python3
"""Tuning aid (GPU box): time r4d_conv1d_f32 on the encoder's GEMM shapes for every tile shape
(R4D_GEMM_TILE forces one; each tile runs in a fresh process because the override is read once)."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [  # (M, K, N, epilogue)
    (8864, 512, 1536, "none"), (8864, 512, 2048, "gelu"), (8864, 512, 512, "residual"), (8864, 2048, 512, "residual"),
    (4096, 512, 1536, "none"), (4096, 2048, 512, "residual"), (2048, 512, 512, "residual"),
    (8864, 768, 2304, "none"), (8864, 3072, 768, "residual"), (16384, 1024, 4096, "none"),
    # bench.py's fused step (G=4 batches): indices 10..13
    (35456, 512, 1536, "none"), (35456, 512, 2048, "gelu"), (35456, 2048, 512, "residual"), (35456, 512, 512, "residual"),
    (8192, 8192, 8192, "none"),            # 14: main-loop asymptote (vendor 154.6 TF here)
]


def child():
    import torch
    sys.path.insert(0, REPO)
    from rag4dyg_amd import ops
    dev = torch.device("cuda:0")
    tile = os.environ.get("R4D_GEMM_KC_TILE" if os.environ.get("R4D_WT") else "R4D_GEMM_TILE", "auto")
    tile = ("kc" + tile) if os.environ.get("R4D_WT") else tile
    sel = os.environ.get("R4D_SHAPES")
    shapes = [SHAPES[int(i)] for i in sel.split(",")] if sel else SHAPES
    if os.environ.get("R4D_SHAPE_LIST"):                 # "MxKxN[:epilogue],..."
        shapes = []
        for t in os.environ["R4D_SHAPE_LIST"].split(","):
            dims, _, epi = t.partition(":")
            shapes.append(tuple(int(v) for v in dims.split("x")) + (epi or "none",))
    for M, K, N, epi in shapes:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(K, N, device=dev) * 0.02
        b = torch.randn(N, device=dev)
        r = torch.randn(M, N, device=dev) if epi == "residual" else None
        wt = w.t().contiguous() if os.environ.get("R4D_WT") else None      # k-contiguous kernel (gemm_f32_kc.hip)
        for _ in range(3):
            ops.conv1d(x, w, b, epi, r, wt)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ops.conv1d(x, w, b, epi, r, wt)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(f"tile={tile:>6} M={M:6d} K={K:5d} N={N:5d} {epi:9s} {ms*1e3:8.1f} us  {2.0*M*K*N/ms/1e9:7.1f} TF", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for t in (sys.argv[1:] or ["0", "1", "2", "3", "auto"]):
            env = dict(os.environ)
            var = "R4D_GEMM_KC_TILE" if env.get("R4D_WT") else "R4D_GEMM_TILE"
            if t != "auto":
                env[var] = t
            else:
                env.pop(var, None)
            subprocess.run([sys.executable, __file__, "child"], env=env, check=True)
