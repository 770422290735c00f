#!/usr/bin/env python3
"""Tuning aid (GPU box): fused vs three-launch attention on the encoder's shapes; every head_dim with an instantiation also the f16x2 kernels
(csrc/attention_h2.hip).  ATT_H2_ONLY=1: that kernel alone (ablated builds: tools/kc_ablate.sh attention_h2.hip ATH_DBG n);
ATT_KBLK=1: K from the key-blocked image (what the encoder runs at d % 256 == 0)."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from rag4dyg_amd import ops
dev = torch.device("cuda:0")
SHAPES = [(128, 128, 2, 256), (128, 277, 2, 256), (128, 128, 6, 128), (128, 300, 6, 128), (128, 128, 8, 96),
          (128, 300, 8, 96), (128, 128, 8, 64), (128, 300, 8, 64), (256, 100, 2, 128)]
if os.environ.get("R4D_SHAPES"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["R4D_SHAPES"].split(",")]
for (B, T, H, hd) in SHAPES:
    d = H * hd
    qkv = torch.randn(B, T, 3 * d, device=dev)
    flop = 2.0 * B * H * T * T * hd
    h2 = ""
    if hd in (32, 64, 96, 128, 256):
        w = ops.pack_h2_words(qkv)
        kb = ops.pack_kblk_words(w, H) if os.environ.get("ATT_KBLK") else None
        run = (lambda: ops.attention_h2_kblk(w, kb, H)) if kb is not None else (lambda: ops.attention_h2(w, H))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e3
        h2 = f"  f16x2{' (key-blocked K)' if kb is not None else ''} {t:8.1f} us ({flop/t/1e6:6.1f} TF)"
    if os.environ.get("ATT_H2_ONLY"):
        print(f"{os.path.basename(os.environ.get('R4D_LIB_PATH', 'product'))} B={B:4d} T={T:4d} H={H} hd={hd:3d}{h2}", flush=True)
        continue
    res = []
    for fused in (1, 2, 0):
        ops.set_attention_fused(fused)
        for _ in range(3):
            ops.attention(qkv, H)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.attention(qkv, H)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    flop = 2.0 * B * H * T * T * hd
    print(f"B={B:4d} T={T:4d} H={H} hd={hd:3d}  fused {res[0]:8.1f} us ({flop/res[0]/1e6:6.1f} TF)  keysplit {res[1]:8.1f} us   unfused {res[2]:8.1f} us ({flop/res[2]/1e6:6.1f} TF){h2}", flush=True)
ops.set_attention_fused(None)
