#!/usr/bin/env python3
"""GPU box: the acceptance table of the f16x2 attention (csrc/attention_h2.hip) -> markdown on stdout.

Per shape and input kind: time per launch of the f16x2 kernel and of the exact-f32 fused kernel (csrc/attention_fused.hip) on the
same q / k / v, and the error of both against a float64 attention of the SAME fp32 inputs (first sequences of the batch; max-norm =
max|d| / max|ref|).  Inputs: N(0,1); logits x 30 (q x 30: a peaked softmax); v x 3e3; an outlier head column x 100 in q, k and v.
The end-to-end checks (G3 / G4 / G12 under gemm mode "f16x2") are tests/test_gpu_pipeline.py and tests/test_gpu_ops.py."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from rag4dyg_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(256, 250, 2, 256), (128, 277, 2, 256), (128, 128, 2, 256), (64, 512, 2, 256), (128, 300, 6, 128), (128, 128, 6, 128), (64, 512, 6, 128)]


def ref64(qkv, H):
    B, T, d3 = qkv.shape
    d = d3 // 3
    hd = d // H
    q, k, v = (t.view(B, T, H, hd).permute(0, 2, 1, 3) for t in qkv.double().split(d, dim=2))
    w = q @ k.transpose(-1, -2) / hd ** 0.5
    w = w.masked_fill(~torch.tril(torch.ones(T, T, dtype=torch.bool, device=qkv.device)), float("-inf"))
    return (torch.softmax(w, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B, T, d)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


def main():
    g = torch.Generator(device=dev).manual_seed(5)
    print("| B x T x H x head_dim | inputs | f16x2: us (row-major K) / us (key-blocked K, what the encoder runs) / TF-equivalent / max-norm error | exact f32: us / TF / max-norm error |")
    print("|---|---|---|---|")
    ops.set_attention_fused(True)
    worse = total = 0
    for B, T, H, hd in SHAPES:
        d = H * hd
        flop = 2.0 * B * H * T * T * hd
        base = torch.randn(B, T, 3 * d, device=dev, generator=g)
        for kind in ("N(0,1)", "logits x 30", "v x 3e3", "outlier column x 100"):
            qkv = base.clone()
            if kind == "logits x 30":
                qkv[:, :, :d] *= 30.0
            elif kind == "v x 3e3":
                qkv[:, :, 2 * d:] *= 3e3
            elif kind.startswith("outlier"):
                qkv[:, :, 5::hd] *= 100.0
            words = ops.pack_h2_words(qkv)
            nb = min(B, 4)
            ref = ref64(qkv[:nb], H)
            kblk = ops.pack_kblk_words(words, H)
            out_h2, out_f32 = ops.attention_h2_kblk(words, kblk, H), ops.attention(qkv, H)
            assert torch.equal(out_h2.view(torch.int32), ops.attention_h2(words, H).view(torch.int32))      # the two K layouts: same bits
            e = [float((o[:nb].double() - ref).abs().max() / ref.abs().max()) for o in (out_h2, out_f32)]
            t = [timed(lambda: ops.attention_h2_kblk(words, kblk, H)), timed(lambda: ops.attention(qkv, H))] if kind == "N(0,1)" else [None, None]
            t_rm = timed(lambda: ops.attention_h2(words, H)) if kind == "N(0,1)" else None
            cell = lambda i: ((f"{t_rm:.1f} / " if i == 0 else "") + f"{t[i]:.1f} / {flop / t[i] / 1e6:.1f} / " if t[i] else ("- / " if i == 0 else "") + "- / - / ") + f"{e[i]:.2e}"
            print(f"| {B} x {T} x {H} x {hd} | {kind} | {cell(0)} | {cell(1)} |", flush=True)
            worse += e[0] > 2.0 * e[1] + 2.5e-7
            total += 1
    ops.set_attention_fused(None)
    print(f"\nf16x2 error above 2 x the exact-f32 kernel's + 2^-22 in {worse} of {total} cases "
          "(q, k, v enter as h2 words: 22 of fp32's 24 significand bits).")


if __name__ == "__main__":
    main()
