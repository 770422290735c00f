#!/usr/bin/env python3
"""Tuning aid (GPU box): phase timeline of pool_scan_ks_kernel from a SCAN_DBG=8 build (tools/kc_ablate.sh score.hip SCAN_DBG 8):
s_memrealtime stamps (100 MHz) per workgroup: start, loads issued, first tile landed, end.
    R4D_ALLOW_ABLATED_LIB=1 R4D_LIB_PATH=$PWD/tools/_bin/librag4dyg_dbg8.so python tools/scan_timeline.py [N] [d]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from rag4dyg_amd import ops  # noqa: E402

N, d = int(sys.argv[1]) if len(sys.argv) > 1 else 12500, int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
q = ops.normalize_rows(torch.randn(32, d, generator=g).to(dev))
p = ops.normalize_rows(torch.randn(N, d, generator=g).to(dev))
for it in range(4):
    _, _, S = ops.score_topk(q, p, 10, want_scores=True)
    torch.cuda.synchronize()
st = S[16:].contiguous().view(-1).view(torch.int64).cpu().numpy()
nwg = 0
rows = []
grid = int(st[10]) if 0 < st[10] < 10**6 else 10**9
while nwg < grid and (nwg + 1) * 16 <= st.size and st[nwg * 16] > 0 and 0 < st[nwg * 16 + 3] - st[nwg * 16] < 10**7:
    rows.append(st[nwg * 16:nwg * 16 + 16])
    nwg += 1
a = np.array(rows, dtype=np.int64)
t0 = a[:, 0].min()
rel = (a[:, :4] - t0) * 10.0      # ns
print(f"N={N} d={d}: {nwg} workgroups; kernel span (first start -> last end) {rel[:, 3].max() / 1e3:.2f} us")
rel = (a - t0) * 10.0
for name, col in (("start", 0), ("loads issued", 1), ("first tile landed", 2), ("tile 0 MFMAs done", 8), ("tile 0 reduced", 9),
                  ("tile 1 MFMAs done", 12), ("tile 1 reduced", 13), ("stores issued", 7), ("end (stores done)", 3)):
    v = rel[:, col] / 1e3
    print(f"  {name:18s} min {v.min():6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f} us")
dur = (a[:, 3] - a[:, 0]) * 0.01
wait = (a[:, 2] - a[:, 1]) * 0.01
print(f"  per-workgroup duration: median {np.median(dur):.2f} max {dur.max():.2f} us; wait for first tile: median {np.median(wait):.2f} max {wait.max():.2f} us")
clk = (a[:, 6] - a[:, 5]) / np.maximum(a[:, 3] - a[:, 0], 1) * 0.1      # shader cycles per 10 ns -> GHz
print(f"  shader clock over the workgroup's lifetime (s_memtime / s_memrealtime): median {np.median(clk):.2f} GHz, min {clk.min():.2f}, max {clk.max():.2f}")
print("  workgroups per XCC id:", np.bincount((a[:, 4] & 0xf).astype(int)))

# per-wavefront stamps (two-tiles-in-flight bf16x3 form): [nwg * KW waves][8]
KW = 8 if d in (512, 768, 1024) else (4 if d >= 128 else d // 32)
ws = st[nwg * 16: nwg * 16 + nwg * KW * 8].reshape(nwg, KW, 8)
if (ws[:, :, 0] > 0).all() and (ws[:, :, 7] >= ws[:, :, 0]).all():
    rel = (ws - t0) * 0.01
    names = ("start", "loads issued", "tile 0 + queries landed", "queries split", "tile 0 MFMAs done", "tile 1 MFMAs done", "past the barrier", "stores issued")
    print("  per wavefront (median over workgroups; us):")
    print("    wave " + "".join(f"{n[:14]:>16s}" for n in names))
    for w in range(KW):
        print(f"    {w:4d} " + "".join(f"{np.median(rel[:, w, c]):16.2f}" for c in range(8)))
    print("    slowest wavefront of a workgroup, median over workgroups: " + ", ".join(f"{names[c]} {np.median(rel[:, :, c].max(axis=1)):.2f}" for c in (2, 4, 5)))
