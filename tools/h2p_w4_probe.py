import os, sys, torch, json
sys.path.insert(0, "/root/repo")
from rag4dyg_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
res = {}
for (M, K, N, epi) in ((63232, 512, 1536, "none"), (63232, 512, 2048, "gelu"), (63232, 2048, 512, "residual"), (63232, 512, 512, "residual"), (1000, 512, 2048, "gelu")):
    x = torch.randn(M, K, generator=g).to(dev); w = (torch.randn(K, N, generator=g) * 0.05).to(dev); b = torch.randn(N, generator=g).to(dev)
    r = torch.randn(M, N, generator=g).to(dev) if epi == "residual" else None
    planes, lines = ops.split2_planes(w), ops.split2_lines(x)
    y = ops.conv1d_h2p(lines, planes, b, epi, r, out_lines=(epi == "gelu"))
    for _ in range(3): ops.conv1d_h2p(lines, planes, b, epi, r, out_lines=(epi == "gelu"))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.conv1d_h2p(lines, planes, b, epi, r, out_lines=(epi == "gelu"))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    res[f"{M}x{K}x{N}:{epi}"] = (round(us, 1), round(2.0 * M * K * N / us / 1e6, 1), int(y.view(torch.int32).to(torch.int64).sum().item()) if y.dtype != torch.int16 else int(y.to(torch.int64).sum().item()))
print(json.dumps({"w4": os.environ.get("R4D_GEMM_H2P_W4", "0"), "res": res}))
