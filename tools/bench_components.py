#!/usr/bin/env python3
"""Secondary measurements (GPU box): the HBM-bound kernels of the path on their own rooflines.

  scan      (S+1)/2 cosine scan at the reference query batch (32) over N pool rows + top-10
  jaccard   f64 Jaccard matrices: real hepth/11 sets (golden fixture) and a synthetic 20k x 20k scale-up
  pool      pool-encode sequences/s (reference batching, fused groups)
Each line: achieved = algorithmic bytes / HIP-event time of the kernel class (r4d_profile_* hooks), peak 8 TB/s.
CPU baselines: oracle single-thread python-set Jaccard (as retrieval_data_annotation.py:36-41) on a bounded sample.
"""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from rag4dyg_amd import _lib, ops, synth                                   # noqa: E402
from bench import host_cores                                                # noqa: E402

PEAK = 8000.0
dev = torch.device("cuda:0")
lib = _lib.load()


def profile(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    lib.r4d_profile_enable(1)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    out = {}
    for c in range(lib.r4d_profile_num_classes()):
        ms, n, w = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        lib.r4d_profile_read(c, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(w))
        if n.value:
            out[lib.r4d_profile_class_name(c).decode()] = dict(us=1e3 * ms.value / n.value, launches=n.value // reps,
                                                              gbs=w.value / (ms.value * 1e-3) / 1e9, work=w.value / n.value)
    lib.r4d_profile_enable(0)
    return wall, out


def emit(**kw):
    print(json.dumps(kw), flush=True)


def graph_wall(fn, iters=20, reps=5):
    """GPU-side time of one `fn()` launch sequence: `iters` back-to-back copies captured into one HIP graph (no Python
    or launch-API time between the kernels, only the GPU's own kernel-to-kernel gaps), replayed `reps` times."""
    fn(); torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        fn()
        st.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(iters):
                fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / (reps * iters)


def scan():
    s3 = ops.gemm_split3_enabled()                      # R4D_GEMM_SPLIT3=0: the exact-f32 form of the scan
    cases = [(12500, 512), (100000, 512), (12500, 768), (100000, 768)]
    if os.environ.get("R4D_SCAN_CASES"):
        cases = [tuple(int(x) for x in c.split("x")) for c in os.environ["R4D_SCAN_CASES"].split(",")]
    for N, d in cases:
        Q, k = 32, 10
        g = torch.Generator().manual_seed(1)
        q = ops.normalize_rows(torch.randn(Q, d, generator=g).to(dev))
        p = ops.normalize_rows(torch.randn(N, d, generator=g).to(dev))
        wall, pr = profile(lambda: ops.score_topk(q, p, k), 50)
        gw = float("nan") if os.environ.get("R4D_NO_GRAPH") else graph_wall(lambda: ops.score_topk(q, p, k))
        s = pr["pool_scan"]
        emit(component="scan", N=N, d=d, Q=Q, k=k, host_loop_wall_us=round(wall * 1e6, 1), gpu_wall_us=round(gw * 1e6, 2),
             kernel="pool_scan", kernel_us=round(s["us"], 2), algorithmic_bytes=s["work"],
             roofline={"bound": "hbm", "achieved": round(s["gbs"], 1), "peak": PEAK, "unit": "GB/s",
                       "frac": round(s["gbs"] / PEAK, 4)},
             topk_us=round(sum(v["us"] * v["launches"] for n, v in pr.items() if n == "topk"), 2),
             topk_launches=sum(v["launches"] for n, v in pr.items() if n == "topk"),
             operands="bf16x3" if s3 else "f32", env={k_: v for k_, v in os.environ.items() if k_.startswith("R4D_SCAN")},
             queries_per_s_scan_topk=round(Q / gw, 1))
    # full-row ranking (file-compat mode) at the north-star pool size
    S = torch.rand(32, 100000, generator=torch.Generator().manual_seed(2)).to(dev)
    wall, pr = profile(lambda: ops.argsort_desc(S), 5)
    emit(component="argsort", rows=32, n=100000, kernel_us=round(pr["argsort"]["us"], 1), wall_us=round(wall * 1e6, 1))


def topk():
    """Selection alone: per-round cost = slope over k."""
    for n in (12500, 100000):
        S = torch.rand(32, n, generator=torch.Generator().manual_seed(3)).to(dev)
        for k in (1, 10, 64):
            gw = graph_wall(lambda: ops.topk_f32(S, k))
            emit(component="topk", rows=32, n=n, k=k, gpu_wall_us=round(gw * 1e6, 2))


def jaccard():
    from oracle import jaccard_ref
    g = np.load(os.path.join(REPO, "tests", "golden", "g5_jaccard_hepth.npz"))
    vocab = len(g["vocab_tokens"])
    t = {k: torch.from_numpy(g[k]).to(dev) for k in ("tr_out_ptr", "tr_out_idx", "tr_in_ptr", "tr_in_idx", "te_out_ptr", "te_out_idx")}
    cases = [("hepth train_out x train_out", "tr_out", "tr_out", True), ("hepth train_in x train_in", "tr_in", "tr_in", True),
             ("hepth test_out x train_out", "te_out", "tr_out", False)]
    for name, a, b, zd in cases:
        na, nb = t[a + "_ptr"].numel() - 1, t[b + "_ptr"].numel() - 1
        wall, pr = profile(lambda: ops.jaccard(t[a + "_ptr"], t[a + "_idx"], t[b + "_ptr"], t[b + "_idx"], vocab, zd), 20)
        j = pr["jaccard"]
        prep = pr.get("jaccard_prep")
        emit(component="jaccard", case=name, pairs=na * nb, kernel_us=round(j["us"], 2), pairs_per_s=round(na * nb / (j["us"] * 1e-6)),
             prep_us=round(prep["us"], 2) if prep else 0.0, call_wall_us=round(wall * 1e6, 1),
             roofline={"bound": "hbm", "achieved": round(j["gbs"], 1), "peak": PEAK, "unit": "GB/s", "frac": round(j["gbs"] / PEAK, 4)})
    # CPU baseline: the reference's single-thread double loop on a bounded sample of the same sets
    lists = [g["tr_out_idx"][g["tr_out_ptr"][i]:g["tr_out_ptr"][i + 1]].tolist() for i in range(len(g["tr_out_ptr"]) - 1)]
    t0 = time.perf_counter()
    jaccard_ref.occurrence_matrix_naive(lists[:600], lists)
    el = time.perf_counter() - t0
    emit(component="jaccard_cpu_baseline", kind="port", cores=1, sample="600 x 3965 hepth out-set pairs, python sets per pair",
         pairs_per_s=round(600 * len(lists) / el))
    # synthetic scale-up: 20,000 out-sets / in-sets, V0 = 11,901
    sh = synth.Shape("reddit_like", 11901, 11, 2, 8, 512, (8, 133, 512), (8, 133, 512))
    for kind, ins in (("out-sets", False), ("in-sets", True)):
        ptr, idx = synth.output_sets(sh, 20000, in_sets=ins)
        P, I = torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev)
        wall, pr = profile(lambda: ops.jaccard(P, I, P, I, sh.v0, True), 5)
        j = pr["jaccard"]
        prep = pr.get("jaccard_prep")
        emit(component="jaccard", case=f"synthetic 20000 x 20000 {kind} (mean size {idx.size / 20000:.1f})", pairs=4e8,
             kernel_us=round(j["us"], 1), pairs_per_s=round(4e8 / (j["us"] * 1e-6)),
             prep_us=round(prep["us"], 2) if prep else 0.0, call_wall_us=round(wall * 1e6, 1),
             roofline={"bound": "hbm", "achieved": round(j["gbs"], 1), "peak": PEAK, "unit": "GB/s", "frac": round(j["gbs"] / PEAK, 4)})


def pool():
    sys.path.insert(0, REPO)
    from bench import build_model
    from rag4dyg_amd.retrieval import encode_batches, right_pad_batches
    for name in ("UCI_13", "wikiv2"):
        shape = synth.SHAPES[name]
        m = build_model(shape, dev)
        seqs = synth.sequences(shape, 12500, "pool", seed=2026)
        b = right_pad_batches(seqs, 32, shape.pad_id, dev)
        encode_batches(m, b[:40]); torch.cuda.synchronize()
        t0 = time.perf_counter()
        encode_batches(m, b)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        emit(component="pool_encode", shape=name, sequences=12500, padded_tokens=int(sum(x.numel() for x in b)),
             seqs_per_s=round(12500 / el, 1))


def generator(shape_name="UCI_13", L=6, H=8, d=768, topk=7, pool_n=512):
    """RAG generator inference (graph-pooling fusion of top-7 retrieved sequences + greedy decode, val
    mode = 11 tokens per query, batch 1 like the reference) on UCI_13-shaped synthetic data, model L6 H8 d768
    (scripts/train_generator/train_rag_graphpooling_UCI_seed.sh), next to the oracle on the host cores."""
    import types
    from oracle import generator_ref, gpt2_ref
    from rag4dyg_amd import generator as gen
    from rag4dyg_amd.gpt2 import GPT2Config, GPT2LMHeadModelRAG
    shape = synth.SHAPES[shape_name]
    V = shape.vocab_generator if shape_name == "reddit" else shape.vocab
    sd = gpt2_ref.make_state_dict(L, d, V, n_positions=1024, seed=4, random_affine=True)
    model = GPT2LMHeadModelRAG(GPT2Config(vocab_size=V, n_positions=1024, n_ctx=1024, n_embd=d, n_layer=L, n_head=H))
    model.load_state_dict(sd, strict=False); model.tie_weights()
    gnn = model.get_gnn(d, d // 2, d, 1, 0.2)
    convs = [(gnn.convs[0].lin.weight.detach().clone(), gnn.convs[0].bias.detach().clone())]
    model = model.to(dev).eval()
    pool_seqs = [s.tolist() for s in synth.sequences(shape, pool_n, "pool", seed=1)]
    queries = [s.tolist() for s in synth.sequences(shape, 256, "query", seed=2)]
    rng = np.random.default_rng(0)
    idxs = [rng.permutation(pool_n)[:topk].tolist() for _ in queries]
    ds = types.SimpleNamespace(retrieval_sources=pool_seqs)
    args = types.SimpleNamespace(fusion="graphpooling", m=1, topK=topk)
    tok = types.SimpleNamespace(encode=lambda s: [shape.v0], pad_token_id=shape.pad_id)     # <|endoftext|> = V0
    eos = tok.encode("")[0]

    def run_gpu(qs):
        n = 0
        for q, ix in qs:
            out = gen.greedy_decode_rag(args, model, tok, ds, q, ix, "val", 1024, 12)
            n += len(out) - len(q)
        torch.cuda.synchronize()
        return n
    run_gpu(list(zip(queries[:4], idxs[:4])))
    t0 = time.perf_counter()
    ntok1 = run_gpu(list(zip(queries[:16], idxs[:16])))
    el1 = time.perf_counter() - t0
    batches = [(queries[b0:b0 + 32], idxs[b0:b0 + 32]) for b0 in range(0, len(queries), 32)]     # --per_gpu_eval_batch_size 32
    list(gen.decode_rag_batches(args, model, tok, ds, batches[:3], "val", 1024, 12))                # warm-up (both decoder slots, graphs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ntok = 0
    for (qs, _ix), outs in zip(batches, gen.decode_rag_batches(args, model, tok, ds, batches, "val", 1024, 12)):
        ntok += sum(len(o) - len(q) for o, q in zip(outs, qs))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # the same queries at --per_gpu_eval_batch_size 128 (the throughput lever of a latency-bound step: same ids per query)
    big = [(queries[b0:b0 + 128], idxs[b0:b0 + 128]) for b0 in range(0, len(queries), 128)]
    list(gen.decode_rag_batches(args, model, tok, ds, big[:2], "val", 1024, 12)); torch.cuda.synchronize()
    t0 = time.perf_counter()
    ntok128 = 0
    for (qs, _ix), outs in zip(big, gen.decode_rag_batches(args, model, tok, ds, big, "val", 1024, 12)):
        ntok128 += sum(len(o) - len(q) for o, q in zip(outs, qs))
    torch.cuda.synchronize()
    el128 = time.perf_counter() - t0
    t0 = time.perf_counter()
    ncpu = 0
    for q, ix in list(zip(queries, idxs))[:3]:
        fn = lambda t, ix=ix: generator_ref.fusion_graphpooling_embeds(sd, pool_seqs, t, ix, topk, convs)
        out = generator_ref.greedy_decode_rag(sd, H, fn, q, eos, "val", 1024, 12)
        ncpu += len(out) - len(q)
    elc = time.perf_counter() - t0
    emit(component="generator_decode", shape=shape_name, model=f"L{L} H{H} d{d} V{V}", pool=pool_n, fusion="graphpooling", topK=topk,
         queries=len(queries), mean_query_len=round(float(np.mean([len(q) for q in queries])), 1), tokens=ntok,
         tokens_per_s=round(ntok / el, 1), queries_per_s=round(len(queries) / el, 2), batch=32,
         tokens_per_s_batch1=round(ntok1 / el1, 1), tokens_per_s_batch128=round(ntok128 / el128, 1),
         cpu_baseline={"kind": "port", "cores": host_cores(), "tokens_per_s": round(ncpu / elc, 2),
                       "sample": "3 queries, oracle torch-CPU fp32 (fusion + full forward per token, as the reference)"})



def simpledyg_eval():
    """SimpleDyG greedy link-prediction decode (utils/Evaluation_SimpleDyG.py:120-145) at BASELINE config 1's model
    (L6 H8 d768 V1800): test-set histories of the UCI_13 length distribution, val mode (11 tokens per query) and test mode
    (until <|endoftext|> or the context limit; random-init weights rarely emit it, so this is the long-generation case), batch 32,
    next to the oracle's one-at-a-time loop on the host cores."""
    import types
    from oracle import gpt2_ref
    from rag4dyg_amd import evaluation
    from rag4dyg_amd.gpt2 import GPT2Config, GPT2LMHeadModel
    shape = synth.SHAPES["UCI_13"]
    L, H, d, V = 6, 8, 768, shape.vocab - 1                    # SimpleDyG omits [MASK] (main_SimpleDyG.py:91-95)
    sd = gpt2_ref.make_state_dict(L, d, V, n_positions=1024, seed=9, random_affine=True)
    model = GPT2LMHeadModel(GPT2Config(vocab_size=V, n_positions=1024, n_ctx=1024, n_embd=d, n_layer=L, n_head=H))
    model.load_state_dict(sd, strict=False); model.tie_weights()
    model = model.to(dev).eval()
    queries = [s.tolist() for s in synth.sequences(shape, 256, "query", seed=3)]
    tok = types.SimpleNamespace(encode=lambda s: [shape.v0], pad_token_id=shape.pad_id)
    out = {}
    for mode, budget in (("val", None), ("test", 160)):
        max_len = 1024 if mode == "val" else None
        evaluation.greedy_decode_batch(model, tok, queries[:32], mode, 1024 if mode == "val" else max(len(q) for q in queries[:32]) + budget, 0, dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ntok = 0
        for b0 in range(0, len(queries), 32):
            qs = queries[b0:b0 + 32]
            ml = 1024 if mode == "val" else max(len(q) for q in qs) + budget      # test mode: a context limit `budget` tokens past the longest prompt
            outs = evaluation.greedy_decode_batch(model, tok, qs, mode, ml, 0, dev)
            ntok += sum(len(o) - len(q) for o, q in zip(outs, qs))
        torch.cuda.synchronize()
        out[mode] = (ntok, time.perf_counter() - t0)
    t0 = time.perf_counter()
    ncpu = 0
    for q in queries[:2]:
        ncpu += len(gpt2_ref.greedy_decode(sd, H, q, shape.v0, "val")) - len(q)
    elc = time.perf_counter() - t0
    emit(component="simpledyg_greedy_eval", shape="UCI_13", model=f"L{L} H{H} d{d} V{V}", queries=len(queries), batch=32,
         val_tokens_per_s=round(out["val"][0] / out["val"][1], 1), val_queries_per_s=round(len(queries) / out["val"][1], 1),
         test_tokens=out["test"][0], test_tokens_per_s=round(out["test"][0] / out["test"][1], 1),
         cpu_baseline={"kind": "port", "cores": host_cores(), "tokens_per_s": round(ncpu / elc, 2),
                       "sample": "2 queries, val mode, oracle torch-CPU fp32 (full forward per token, as the reference)"})


def generator_reddit():
    """BASELINE config 5 shape: reddit t=11, L2 H8 d512, V = 11,919, pool 10,527, top-7 (the script) and top-5 (BASELINE.json)."""
    generator("reddit", 2, 8, 512, 7, 10527)
    generator("reddit", 2, 8, 512, 5, 10527)


_JAC_CPU = r"""
import json, os, sys, time, multiprocessing as mp
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import jaccard_ref
cores = int(sys.argv[2])
g = np.load(os.path.join(sys.argv[1], "tests", "golden", "g5_jaccard_hepth.npz"))
lists = [g["tr_out_idx"][g["tr_out_ptr"][i]:g["tr_out_ptr"][i + 1]].tolist() for i in range(len(g["tr_out_ptr"]) - 1)]
rows = 150 * cores
blocks = [lists[i:i + 150] for i in range(0, rows, 150)]
t0 = time.perf_counter()
with mp.get_context("fork").Pool(cores) as pool:
    pool.starmap(jaccard_ref.occurrence_matrix_naive, [(b_, lists) for b_ in blocks])
el = time.perf_counter() - t0
print(json.dumps({"rows": rows, "cols": len(lists), "seconds": el}))
"""


def training():
    """one retriever training iteration (five forwards with saved activations, losses, encoder backward, clip,
    AdamW) at the UCI_13 script's shape (L4 H2 d512, batch 64: scripts/train_retriever/train_retriever_UCI_13.sh:8-12) on
    synthetic UCI_13-length sequences.  flop = 3 x the forward's algorithmic encoder flop (forward + dX + dW) of the five
    padded batches; wall clock includes the host side of the step (aug, loss autograd over [5, B, d], ~40 optimizer launches)."""
    import random
    import types
    from bench import build_model, f_enc
    from rag4dyg_amd import training as tr
    from rag4dyg_amd.retrieval import right_pad_batches
    shape = synth.SHAPES["UCI_13"]
    B = 64
    m = build_model(shape, dev)
    m.config.eta, m.config.gamma = 0.8, 0.4
    nb = 22
    seqs = synth.sequences(shape, 3 * B * nb, "query", seed=11)
    args = types.SimpleNamespace(device=dev, temperature=0.1, lambda_decay=1e-4, alpha=1.0, per_gpu_train_batch_size=B,
                                 max_grad_norm=1.0, gradient_accumulation_steps=1)
    times = (torch.rand(3 * B * nb) * 1e4).to(dev)
    batches = []
    for i in range(nb):
        trip = [right_pad_batches(seqs[(3 * i + j) * B:(3 * i + j + 1) * B], B, shape.pad_id, "cpu")[0] for j in range(3)]   # as the DataLoader hands them over
        idx = [torch.arange((3 * i + j) * B, (3 * i + j + 1) * B).view(B, 1) for j in range(3)]
        batches.append((*trip, *idx))
    n = nb - 2
    flop = sum(3.0 * (f_enc(shape, B, b[0].shape[1]) * 3 + f_enc(shape, B, b[1].shape[1]) + f_enc(shape, B, b[2].shape[1]))
               for b in batches[2:])
    wall = {}
    for mode in ("warm", "eval", "train"):               # "train": model.train(), dropout 0.1 at the four sites like the reference
        m.train(mode == "train")
        trainer = tr.EncoderTrainer(m, seed=42)
        opt = tr.AdamW(trainer.params, trainer.grads, lr=1e-5, eps=1e-8, weight_decay=0.0, flat_grads=trainer.flat_grads)
        random.seed(0)
        for b in batches[:2]:
            tr.training_step(args, m, trainer, opt, b, times)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in batches[2:]:
            tr.training_step(args, m, trainer, opt, b, times, sync=False)      # as train_epoch runs it: losses summed on the device
        torch.cuda.synchronize()
        wall[mode] = time.perf_counter() - t0
    el = wall["train"]
    # device-side split of one step (forward / backward / optimizer) with events
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    b = batches[2]
    a1, a2 = tr.aug(b[0], 0.8, 0.4, m.config.vocab_size - 1)
    torch.cuda.synchronize()
    ev[0].record()
    emb = trainer.forward([x.to(dev) for x in (b[0], b[1], b[2], a1, a2)])
    ev[1].record()
    trainer.backward(torch.randn_like(emb))
    ev[2].record()
    opt.step(1.0)
    ev[3].record()
    torch.cuda.synchronize()
    emit(component="retriever_training_step", shape="UCI_13", batch=B, steps=n, dropout=0.1, ms_per_step=round(1e3 * el / n, 3),
         sequences_per_s=round(5 * B * n / el, 1), tflops=round(flop / el / 1e12, 1),
         ms_per_step_dropout_off=round(1e3 * wall["eval"] / n, 3),
         forward_ms=round(ev[0].elapsed_time(ev[1]), 3), backward_ms=round(ev[1].elapsed_time(ev[2]), 3),
         optimizer_ms=round(ev[2].elapsed_time(ev[3]), 3), padded_T=[int(x.shape[1]) for x in b[:3]])


def training_cpu():
    """The same training iteration on the host cores: the oracle's grad-enabled forward + torch CPU autograd + the restated
    AdamW (oracle/train_ref.py), ONE step at batch 16 of the same sequence distribution (bounded sample)."""
    import random
    from oracle import gpt2_ref, train_ref
    shape = synth.SHAPES["UCI_13"]
    B = 16
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = gpt2_ref.make_state_dict(shape.n_layer, shape.n_embd, shape.vocab, n_positions=1024, seed=3)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    seqs = synth.sequences(shape, 3 * B, "query", seed=11)

    def pad(chunk):
        T = max(len(e) for e in chunk)
        b = torch.full((len(chunk), T), int(shape.pad_id), dtype=torch.int64)
        for i, e in enumerate(chunk):
            b[i, :len(e)] = torch.as_tensor(e)
        return b
    a, p_, n_ = pad(seqs[:B]), pad(seqs[B:2 * B]), pad(seqs[2 * B:])
    idx = torch.arange(3 * B).view(3, B).t().contiguous()
    t0 = time.perf_counter()
    r = train_ref.training_step(sd, shape.n_head, a, p_, n_, torch.rand(3 * B) * 1e4, idx, 0.8, 0.4, 1.0, 0.1, 1e-4,
                                shape.vocab - 1, 0, with_grad=True)
    r["loss"].backward()
    names = [k for k in sd if sd[k].grad is not None]
    coef, _ = train_ref.clip_coefficient([sd[k].grad for k in names], 1.0)
    for k in names:
        train_ref.adamw_step(sd[k].detach(), sd[k].grad * coef, torch.zeros_like(sd[k]), torch.zeros_like(sd[k]), 1, 1e-5)
    el = time.perf_counter() - t0
    emit(component="retriever_training_step_cpu", kind="port", cores=cores, shape="UCI_13", batch=B,
         sample=f"one step, batch {B} (5 x {B} sequences), padded T = {[int(a.shape[1]), int(p_.shape[1]), int(n_.shape[1])]}",
         seconds=round(el, 2), sequences_per_s=round(5 * B / el, 2))


def jaccard_cpu_all_cores():
    """the reference's python-set double loop (retrieval_data_annotation.py:36-41) on ALL host cores (row blocks in
    a process pool, started from a fresh interpreter that never touches the GPU) next to the single-thread line of jaccard()."""
    import subprocess
    cores = host_cores()
    out = subprocess.run([sys.executable, "-c", _JAC_CPU, REPO, str(cores)], capture_output=True, text=True, check=True).stdout
    r = json.loads(out.strip().splitlines()[-1])
    emit(component="jaccard_cpu_baseline", kind="port", cores=cores,
         sample=f"{r['rows']} x {r['cols']} hepth out-set pairs, python sets per pair, {cores} processes",
         pairs_per_s=round(r["rows"] * r["cols"] / r["seconds"]))


if __name__ == "__main__":
    for part in (sys.argv[1:] or ["scan", "topk", "jaccard", "jaccard_cpu", "pool", "generator", "generator_reddit", "simpledyg", "training", "training_cpu"]):
        {"scan": scan, "topk": topk, "jaccard": jaccard, "jaccard_cpu": jaccard_cpu_all_cores, "pool": pool, "generator": generator,
         "generator_reddit": generator_reddit, "simpledyg": simpledyg_eval, "training": training, "training_cpu": training_cpu}[part]()
