import sys, torch
sys.path.insert(0, '.')
from rag4dyg_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
for (M, K, N) in [(128, 32, 256), (130, 64, 100), (257, 512, 513), (1000, 96, 1536), (40000, 64, 300), (33000, 512, 512), (20000, 128, 1000)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(K, N, device=dev); b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev)
    pl = ops.split3_planes(w)
    pl2 = ops.split3_planes(w.t().contiguous(), transposed=True)
    assert torch.equal(pl, pl2)
    p = pl.view(torch.bfloat16).double()
    assert torch.equal(p.sum(0), w.t().double()), "planes not exact"
    for epi in ("none", "gelu", "residual"):
        y = ops.conv1d_s3(x, pl, b, epi, r if epi == "residual" else None)
        ref = x.double() @ w.double() + b.double()
        if epi == "gelu": ref = torch.nn.functional.gelu(ref, approximate='tanh')
        if epi == "residual": ref = ref + r.double()
        y0 = ops.conv1d(x, w, b, epi, r if epi == "residual" else None, w.t().contiguous())
        e = float((y.double() - ref).abs().max() / ref.abs().max()); e0 = float((y0.double() - ref).abs().max() / ref.abs().max())
        print(M, K, N, epi, "s3 err %.2e  f32 err %.2e" % (e, e0), flush=True)
        assert e < 2e-6, e
print("ok")
