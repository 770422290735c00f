#!/usr/bin/env python3
"""Tuning aid (GPU box): the bf16x3 GEMM (r4d_conv1d_s3_f32) against the exact-f32 MFMA kernel (r4d_conv1d_f32 with a
k-contiguous weight copy) on the encoder's shapes -- time per launch, effective TFLOP/s, and the error of BOTH against a
float64 product (max-norm and element-wise).  Each forced tile (R4D_GEMM_S3_TILE) runs in a fresh process.

    python tools/s3_bench.py [tile ...]        # default: 0 1 auto
"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [  # (M, K, N, epilogue): bench.py's fused step (8 batches of 32 x T): M ~ 63k; wikiv2 d = 768
    (63232, 512, 1536, "none"), (63232, 512, 2048, "gelu"), (63232, 2048, 512, "residual"), (63232, 512, 512, "residual"),
    (8864, 512, 1536, "none"), (8864, 2048, 512, "residual"),
    (40960, 768, 2304, "none"), (40960, 3072, 768, "residual"),
    (8192, 8192, 8192, "none"),
]


def child():
    import torch
    sys.path.insert(0, REPO)
    from rag4dyg_amd import ops
    dev = torch.device("cuda:0")
    tile = os.environ.get("R4D_GEMM_S3_TILE", "auto")
    sel = os.environ.get("R4D_SHAPES")
    shapes = [SHAPES[int(i)] for i in sel.split(",")] if sel else SHAPES
    g = torch.Generator(device=dev).manual_seed(1)
    for M, K, N, epi in shapes:
        x = torch.randn(M, K, device=dev, generator=g)
        w = torch.randn(K, N, device=dev, generator=g) * 0.02
        b = torch.randn(N, device=dev, generator=g)
        r = torch.randn(M, N, device=dev, generator=g) if epi == "residual" else None
        wt = w.t().contiguous()
        planes = ops.split3_planes(w)
        planes2 = ops.split2_planes(w)
        if os.environ.get("S3_TIME_ONLY"):                     # ablated builds (tools/kc_ablate.sh gemm_s3.hip S3_DBG n): timing only
            kind = os.environ.get("S3_BENCH_KIND")
            if kind == "h2p":
                lines_ = ops.split2_lines(x)
                fn = lambda: ops.conv1d_h2p(lines_, planes2, b, epi, r, out_lines=(epi == "gelu"))      # as the encoder launches it
            else:
                fn = (lambda: ops.conv1d_h2(x, planes2, b, epi, r)) if kind == "h2" else (lambda: ops.conv1d_s3(x, planes, b, epi, r))
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20 if M * K * N < 2e11 else 5
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            print(json.dumps({"lib": os.path.basename(os.environ.get("R4D_LIB_PATH", "product")), "tile": tile, "M": M, "K": K, "N": N,
                              "epi": epi, "s3_us": round(ms * 1e3, 1), "s3_TF": round(2.0 * M * K * N / ms / 1e9, 1)}), flush=True)
            continue
        rows = min(M, 4096)                                    # float64 reference on a slice (all columns)
        ref = x[:rows].double() @ w.double() + b.double()
        if epi == "gelu":
            ref = 0.5 * ref * (1.0 + torch.tanh(0.7978845608028654 * (ref + 0.044715 * ref ** 3)))
        elif epi == "residual":
            ref = ref + r[:rows].double()
        out = {}
        lines = ops.split2_lines(x) if K % 32 == 0 else None          # A as f16x2 lines: the LDS-DMA form (gemm_h2p.hip), same bits as "h2"
        for name, fn in (("h2p", (lambda: ops.conv1d_h2p(lines, planes2, b, epi, r)) if lines is not None else None),
                         ("h2", lambda: ops.conv1d_h2(x, planes2, b, epi, r)), ("s3", lambda: ops.conv1d_s3(x, planes, b, epi, r)),
                         ("f32", lambda: ops.conv1d(x, w, b, epi, r, wt))):
            if fn is None:
                continue
            y = fn()
            d = (y[:rows].double() - ref).abs()
            out[name + "_maxnorm"] = float(d.max() / ref.abs().max())
            out[name + "_rms"] = float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20 if M * K * N < 2e11 else 5
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            out[name + "_us"] = round(ms * 1e3, 1)
            out[name + "_TF"] = round(2.0 * M * K * N / ms / 1e9, 1)
        if K <= 4096:                                          # the planes are an exact split: hi + mid + lo == w
            p = planes.view(torch.bfloat16).float()
            out["planes_exact"] = bool(torch.equal(p[0].double() + p[1].double() + p[2].double(), wt.double()))
            p2 = planes2.view(torch.float16).double()                  # lines [N, K/32, 2, 32]: hi, lo' of 32 consecutive k
            hi2, lo2 = p2[:, :, 0].reshape(N, K), p2[:, :, 1].reshape(N, K)
            out["h2_planes_relerr"] = float(((hi2 + lo2 / 2048.0 - wt.double()).abs() / wt.double().abs().clamp_min(1e-30)).max())
        print(json.dumps({"tile": tile, "M": M, "K": K, "N": N, "epi": epi, **out}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for t in (sys.argv[1:] or ["0", "1", "auto"]):
            env = dict(os.environ)
            if t != "auto":
                env["R4D_GEMM_S3_TILE"] = t
            else:
                env.pop("R4D_GEMM_S3_TILE", None)
            subprocess.run([sys.executable, __file__, "child"], env=env, check=True)
