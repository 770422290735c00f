"""Where a batch of the RAG generator's evaluation goes (UCI_13 shape, L6 H8 d768, top-7 graph pooling, val mode):
fusion rows / prefill / decode steps, the decode step per kernel class (launch profiler), graph vs kernel-by-kernel.
    python tools/decode_phases.py            (R4D_DECODE_GRAPH=0: no HIP graph)"""
import sys, time, types, os
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from oracle import gpt2_ref
from rag4dyg_amd import generator as gen, synth, gpt2
from rag4dyg_amd.gpt2 import GPT2Config, GPT2LMHeadModelRAG
dev = torch.device("cuda:0")
shape = synth.SHAPES["UCI_13"]
L, H, d, topk = 6, 8, 768, 7
V = shape.vocab
sd = gpt2_ref.make_state_dict(L, d, V, n_positions=1024, seed=4, random_affine=True)
model = GPT2LMHeadModelRAG(GPT2Config(vocab_size=V, n_positions=1024, n_ctx=1024, n_embd=d, n_layer=L, n_head=H))
model.load_state_dict(sd, strict=False); model.tie_weights()
model.get_gnn(d, d // 2, d, 1, 0.2)
model = model.to(dev).eval()
pool_seqs = [s.tolist() for s in synth.sequences(shape, 512, "pool", seed=1)]
queries = [s.tolist() for s in synth.sequences(shape, 256, "query", seed=2)]
rng = np.random.default_rng(0)
idxs = [rng.permutation(512)[:topk].tolist() for _ in queries]
ds = types.SimpleNamespace(retrieval_sources=pool_seqs)
args = types.SimpleNamespace(fusion="graphpooling", m=1, topK=topk)
tok = types.SimpleNamespace(encode=lambda s: [shape.v0], pad_token_id=shape.pad_id)
T = {}
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0
        return r
    return w
gen.fusion_rows_batch = timed("fusion_rows_batch", gen.fusion_rows_batch)
tr = model.transformer
tr.prefill_last = timed("prefill_last", tr.prefill_last)
gpt2.GreedyDecoder.run = timed("decoder.run", gpt2.GreedyDecoder.run)
gpt2.GreedyDecoder._steps = timed("  decoder._steps", gpt2.GreedyDecoder._steps)
for rep in range(2):
    T.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b0 in range(0, 256, 32):
        gen.greedy_decode_rag_batch(args, model, tok, ds, queries[b0:b0 + 32], idxs[b0:b0 + 32], "val", 1024, 12)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print("graph" if os.environ.get("R4D_DECODE_GRAPH", "1") != "0" else "eager", "rep", rep, "total ms/batch", round(el / 8 * 1e3, 2),
          {k: round(v / 8 * 1e3, 2) for k, v in T.items()})
import ctypes
from rag4dyg_amd import _lib
lib = _lib.load()
os.environ["R4D_DECODE_GRAPH"] = "0"
tr.__dict__.pop("_greedy_decoders", None)
gen.greedy_decode_rag_batch(args, model, tok, ds, queries[:32], idxs[:32], "val", 1024, 12)
dec = list(tr._greedy_decoders.values())[0]
dec.use_graph = False
torch.cuda.synchronize()
lib.r4d_profile_enable(1)
dec.active.fill_(0)
gpt2.GreedyDecoder._steps.__wrapped__ if hasattr(gpt2.GreedyDecoder._steps, "__wrapped__") else None
dec._steps(10)
torch.cuda.synchronize()
tot = 0
for c in range(lib.r4d_profile_num_classes()):
    ms, n, wk = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
    lib.r4d_profile_read(c, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(wk))
    if n.value:
        tot += ms.value
        print("   ", lib.r4d_profile_class_name(c).decode(), n.value / 10, "launches/step", round(1e3 * ms.value / 10, 1), "us/step", round(1e3 * ms.value / n.value, 2), "us each")
print("    total us/step", round(1e3 * tot / 10, 1))
lib.r4d_profile_enable(0)
raw = gpt2.GreedyDecoder._steps
for mode in (True, False):
    dec.use_graph = mode
    dec._steps(5); torch.cuda.synchronize()
    t0 = time.perf_counter(); dec._steps(50); torch.cuda.synchronize()
    print("steady-state step wall us", "graph" if mode else "eager", round((time.perf_counter() - t0) / 50 * 1e6, 1))
dec.use_graph = False
gpt2.GreedyDecoder.use_graph = False
os.environ["R4D_DECODE_GRAPH"] = "0"
tr.__dict__.pop("_greedy_decoders", None)
gen.greedy_decode_rag_batch(args, model, tok, ds, queries[:32], idxs[:32], "val", 1024, 12)
torch.cuda.synchronize()
lib.r4d_profile_enable(1)
gen.greedy_decode_rag_batch(args, model, tok, ds, queries[32:64], idxs[32:64], "val", 1024, 12)
torch.cuda.synchronize()
print("real batch (prefill + 11 steps), per class:")
for c in range(lib.r4d_profile_num_classes()):
    ms, n, wk = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
    lib.r4d_profile_read(c, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(wk))
    if n.value:
        print("   ", lib.r4d_profile_class_name(c).decode(), n.value, "launches", round(1e3 * ms.value, 1), "us total", round(1e3 * ms.value / n.value, 2), "us each")
lib.r4d_profile_enable(0)
d0 = list(tr._greedy_decoders.values())[0]
print("t_cap", d0.t_cap, "lens", d0.lens.tolist())
