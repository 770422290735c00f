#!/usr/bin/env python3
"""GPU box: the acceptance table of the split GEMMs (f16x2: csrc/gemm_h2.hip; bf16x3: csrc/gemm_s3.hip) -> markdown on stdout.

For every Conv1D shape of the golden configurations (G2 op shapes, G3 cfg1 / cfg2 / cfg4 / hepth, the bench's fused step)
and every epilogue: error of the f16x2 kernel (r4d_conv1d_h2_f32), of the bf16x3 kernel (r4d_conv1d_s3_f32) and of the exact-f32
MFMA kernel (r4d_conv1d_f32 with the k-contiguous weight copy) against a float64 product of the SAME fp32 inputs -- max-norm (max|d| / max|ref|), rms-relative,
and the element-wise ratio max |d| / (1e-4 |ref| + 1e-5 max|ref|) the parity tests use (pass < 1).  Inputs: N(0,1)
activations, N(0, 0.02) weights (the reference init), N(0,1) biases / residuals; plus one "trained-like" case with outlier
channels (x * 30 on 4 columns), one with a large common offset (x + 50), and -- the edges of the f16x2 operand range -- one with
activations of 1e-3 and one of 3e4.  The end-to-end check (UCI_13 real data, 110 of
110 top-10 lists identical to the reference) is tests/test_gpu_pipeline.py under both settings of ops.set_gemm_split3."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from rag4dyg_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
CASES = [  # (label, M, K, N)
    ("G2 conv1d small", 45, 96, 80), ("G3 cfg2 c_attn (32x128 rows)", 4096, 512, 1536), ("G3 cfg2 c_fc", 4096, 512, 2048),
    ("G3 cfg2 mlp c_proj", 4096, 2048, 512), ("G3 cfg2 attn c_proj", 4096, 512, 512),
    ("G3 cfg1 c_attn (d 768)", 1024, 768, 2304), ("G3 cfg1 mlp c_proj", 1024, 3072, 768), ("G3 hepth c_fc (d 128)", 2048, 128, 512),
    ("bench step c_attn", 63232, 512, 1536), ("bench step c_fc", 63232, 512, 2048), ("bench step mlp c_proj", 63232, 2048, 512),
    ("wikiv2 step mlp c_proj", 40960, 3072, 768),
]


def errs(y, ref):
    d = (y.double() - ref).abs()
    return (float(d.max() / ref.abs().max()), float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()),
            float((d / (1e-4 * ref.abs() + 1e-5 * ref.abs().max())).max()))


def main():
    g = torch.Generator(device=dev).manual_seed(3)
    print("| shape (M x K x N) | epilogue | inputs | f16x2: max-norm / rms / element-wise ratio | bf16x3: max-norm / rms / element-wise ratio | exact f32: max-norm / rms / element-wise ratio |")
    print("|---|---|---|---|---|---|")
    worst = {"h2": 0.0, "s3": 0.0, "f32": 0.0}
    wins = wins2 = wins2b = total = 0
    for label, M, K, N in CASES:
        for kind in ("N(0,1)", "outlier channels", "offset +50", "x 1e-3", "x 3e4") if M <= 4096 else ("N(0,1)",):
            x = torch.randn(M, K, device=dev, generator=g)
            if kind == "outlier channels":
                x[:, :4] *= 30.0
            elif kind == "offset +50":
                x += 50.0
            elif kind == "x 1e-3":
                x *= 1e-3
            elif kind == "x 3e4":
                x *= 3e4
            w = torch.randn(K, N, device=dev, generator=g) * 0.02
            sc = {"x 1e-3": 1e-3, "x 3e4": 3e4}.get(kind, 1.0)
            b = torch.randn(N, device=dev, generator=g) * sc
            r = torch.randn(M, N, device=dev, generator=g) * sc
            planes, planes2, wt = ops.split3_planes(w), ops.split2_planes(w), w.t().contiguous()
            rows = min(M, 4096)
            base = x[:rows].double() @ w.double() + b.double()
            for epi in ("none", "gelu", "residual"):
                ref = base
                if epi == "gelu":
                    ref = 0.5 * base * (1.0 + torch.tanh(0.7978845608028654 * (base + 0.044715 * base ** 3)))
                elif epi == "residual":
                    ref = base + r[:rows].double()
                e2 = errs(ops.conv1d_h2(x, planes2, b, epi, r if epi == "residual" else None)[:rows], ref)
                e3 = errs(ops.conv1d_s3(x, planes, b, epi, r if epi == "residual" else None)[:rows], ref)
                e1 = errs(ops.conv1d(x, w, b, epi, r if epi == "residual" else None, wt)[:rows], ref)
                worst["h2"], worst["s3"], worst["f32"] = max(worst["h2"], e2[2]), max(worst["s3"], e3[2]), max(worst["f32"], e1[2])
                wins += e3[0] <= e1[0]
                wins2 += e2[0] <= e1[0]
                wins2b += e2[0] <= e3[0]
                total += 1
                print(f"| {label}: {M} x {K} x {N} | {epi} | {kind} | {e2[0]:.2e} / {e2[1]:.2e} / {e2[2]:.4f} | {e3[0]:.2e} / {e3[1]:.2e} / {e3[2]:.4f} | {e1[0]:.2e} / {e1[1]:.2e} / {e1[2]:.4f} |")
    print(f"\nf16x2 max-norm error <= exact-f32's in {wins2} of {total} cases and <= bf16x3's in {wins2b}; bf16x3 <= exact-f32's in {wins}; "
          f"worst element-wise ratio (pass < 1): f16x2 {worst['h2']:.4f}, bf16x3 {worst['s3']:.4f}, exact f32 {worst['f32']:.4f}.")


if __name__ == "__main__":
    main()
