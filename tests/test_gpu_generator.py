"""GPU tests of the generator / evaluation rows against vectors produced by the REFERENCE's own code
(tests/golden/g7_generator.npz: fusion_mlp, the greedy loops, at a tiny shape and at BASELINE config 5 = reddit) and,
for the one piece without a reference fixture (GCNConv, torch_geometric absent), against the oracle."""
import types

import numpy as np
import pytest
import torch

from conftest import assert_tokens_equal_or_tie, elementwise_err, load_golden, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _unrag(flat, off):
    return [flat[off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]


class _Tok:
    def __init__(self, pad, eos):
        self.pad_token_id, self.eos_token_id = pad, eos

    def encode(self, text):
        assert text == "<|endoftext|>"
        return [self.eos_token_id]


def _rag_model(sd, L, H, d, V, dev):
    from rag4dyg_amd.gpt2 import GPT2Config, GPT2LMHeadModelRAG
    m = GPT2LMHeadModelRAG(GPT2Config(vocab_size=V, n_positions=1024, n_ctx=1024, n_embd=d, n_layer=L, n_head=H))
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected
    return m


@pytest.mark.parametrize("tag", ["fmlp_tiny", "fmlp_reddit", "fmlp_reddit_real"])
def test_fusion_mlp_logits_and_greedy_ids_equal_reference(dev, tag):
    """utils/model.py:105-164 + Evaluation_generator.py:153-175 as the REFERENCE computed them on CPU: first-step logits
    within 1e-4 (element-wise), generated ids identical for the reference-structured loop (batch 1, full forward per
    token) AND for the batched key/value-cached decode.  fmlp_reddit = BASELINE config 5 shape (L2 H8 d512 V11919, pool
    10,527, top-7) on synthetic ids; fmlp_reddit_real = the same shape on REAL reddit/11 ids (g12: the training lines the
    reference's csv2resources.py regenerates as demonstrations, six of the validation queries it writes)."""
    from oracle import generator_ref, gpt2_ref
    from rag4dyg_amd import generator, ops
    g = load_golden("g12_reddit_generator" if tag.endswith("_real") else "g7_generator")
    L, H, d, V, pad, eos, m, nl, topk, seed = (int(x) for x in g[tag + "_cfg"])
    sd = gpt2_ref.make_state_dict(L, d, V, n_positions=1024, seed=seed, random_affine=True)
    model = _rag_model(sd, L, H, d, V, dev)
    mlp_sd = generator_ref.make_mlp_state(seed + 1, 512, m, nl)
    model.get_mlp(512, m, nl).load_state_dict(mlp_sd)
    model = model.to(dev).eval()
    src = dict(zip(g[tag + "_src_ids"].tolist(), _unrag(g[tag + "_src_flat"], g[tag + "_src_off"])))
    ds = types.SimpleNamespace(retrieval_sources=src)
    args = types.SimpleNamespace(fusion="mlp", m=m, topK=topk)
    tok = _Tok(pad, eos)
    queries, idxs = _unrag(g[tag + "_q_flat"], g[tag + "_q_off"]), g[tag + "_idxs"].tolist()
    want = _unrag(g[tag + "_gen_flat"], g[tag + "_gen_off"])
    layers = generator_ref.mlp_layers_of(mlp_sd)
    for qi in range(2):                                                   # committed first-step logit rows
        rows = generator.fusion_rows(args, model, tok, ds, idxs[qi], topk)
        wte = model.transformer.wte.weight
        Hq = wte[torch.tensor(queries[qi], device=dev)]
        H_aug = torch.cat([Hq[:2], rows, Hq[2:]], dim=0).unsqueeze(0).contiguous()
        hidden = model.transformer.encode(None, H_aug, want_hidden=True)["hidden"]
        lg = ops.lm_logits(hidden[:, -1, :].contiguous(), model.lm_head.weight)[0].cpu().numpy()
        ref = g[tag + "_first_logits"][qi]
        assert rel_err(lg, ref) < TOL and elementwise_err(lg, ref) < 1, (rel_err(lg, ref), elementwise_err(lg, ref))
    exact = 0
    for qi, (q, ix) in enumerate(zip(queries, idxs)):
        def logits_at(prefix, q=q, ix=ix):
            aug = generator_ref.fusion_mlp_embeds(sd, src, q + list(prefix), ix, topk, m, layers, pad)
            return gpt2_ref.gpt2_forward(sd, None, H, inputs_embeds=aug)["logits"][0, -1].numpy()
        got = generator.greedy_decode_rag(args, model, tok, ds, q, ix, "val", 1024, 19)[len(q):]
        exact += assert_tokens_equal_or_tie(got, want[qi], logits_at, f"{tag} query {qi} (batch-1 loop)")
    many = generator.greedy_decode_rag_batch(args, model, tok, ds, queries, idxs, "val", 1024, 19)
    for qi, (q, ix) in enumerate(zip(queries, idxs)):
        def logits_at(prefix, q=q, ix=ix):
            aug = generator_ref.fusion_mlp_embeds(sd, src, q + list(prefix), ix, topk, m, layers, pad)
            return gpt2_ref.gpt2_forward(sd, None, H, inputs_embeds=aug)["logits"][0, -1].numpy()
        exact += assert_tokens_equal_or_tie(many[qi][len(q):], want[qi], logits_at, f"{tag} query {qi} (cached batch)")
    print(f"{tag}: {exact} of {2 * len(queries)} generated lists identical to the reference's")


@pytest.mark.parametrize("tag", ["sdg_tiny", "sdg_cfg1"])
def test_simpledyg_greedy_ids_equal_reference(dev, tag):
    """Evaluation_SimpleDyG.py:126-145 as the REFERENCE model decoded on CPU (val: 11 tokens, test: until the length cap
    / end-of-text; sdg_cfg1 = BASELINE config 1 shape L6 H8 d768): the batched, key/value-cached decode generates the
    same ids; the reference-structured batch-1 loop too (val mode)."""
    from oracle import gpt2_ref
    from rag4dyg_amd.evaluation import greedy_decode, greedy_decode_batch
    from rag4dyg_amd.gpt2 import GPT2Config, GPT2LMHeadModel
    g = load_golden("g7_generator")
    L, H, d, V, eos, seed, max_len, n_spl = (int(x) for x in g[tag + "_cfg"])
    sd = gpt2_ref.make_state_dict(L, d, V, n_positions=1024, seed=seed, random_affine=True)
    m = GPT2LMHeadModel(GPT2Config(vocab_size=V, n_positions=1024, n_ctx=1024, n_embd=d, n_layer=L, n_head=H))
    m.load_state_dict(sd, strict=False)
    m = m.to(dev).eval()
    tok = _Tok(V - 1, eos)
    prompts = _unrag(g[tag + "_p_flat"], g[tag + "_p_off"])

    def logits_for(pr):
        return lambda prefix: gpt2_ref.gpt2_forward(sd, torch.tensor([pr + list(prefix)]), H)["logits"][0, -1].numpy()
    exact = total = 0
    for mode in ("val", "test"):
        want = _unrag(g[f"{tag}_{mode}_flat"], g[f"{tag}_{mode}_off"])
        many = greedy_decode_batch(m, tok, prompts, mode, max_len, n_spl, dev)
        for pr, got, w in zip(prompts, many, want):
            exact += assert_tokens_equal_or_tie(got[len(pr):], w, logits_for(pr), f"{tag} {mode} (cached batch)"); total += 1
        if mode == "val":
            for pr, w in zip(prompts, want):
                got = greedy_decode(m, tok, pr, mode, max_len, n_spl, dev)
                exact += assert_tokens_equal_or_tie(got[len(pr):], w, logits_for(pr), f"{tag} {mode} (batch-1 loop)"); total += 1
    print(f"{tag}: {exact} of {total} generated lists identical to the reference's")


@pytest.mark.parametrize("topk", [7, 5])
def test_reddit_shape_graphpooling_decode_batch32(dev, topk):
    """BASELINE config 5 as written (scripts/train_generator/train_rag_graphpooling_reddit_seed.sh:6-14): L2 H8 d512,
    V = 11,919, pool of 10,527 synthetic sequences, graph-pooling fusion of the top-K retrieved sequences (K = 7 as the
    script runs it, 5 as BASELINE.json words it), batch 32, val mode.  Checked against the oracle: its GPT-2 forward,
    splice and loop are pinned by reference fixtures; its GCNConv restates torch_geometric's published formula (the one
    UNPINNED piece: torch_geometric is neither in the reference tree nor installed)."""
    from oracle import generator_ref, gpt2_ref
    from rag4dyg_amd import generator, synth
    sh = synth.SHAPES["reddit"]
    L, H, d, V = sh.n_layer, sh.n_head, sh.n_embd, sh.vocab_generator
    assert (L, H, d, V) == (2, 8, 512, 11919)
    sd = gpt2_ref.make_state_dict(L, d, V, n_positions=1024, seed=55, random_affine=True)
    model = _rag_model(sd, L, H, d, V, dev)
    gnn = model.get_gnn(d, d // 2, d, 1, 0.2)
    gg = torch.Generator().manual_seed(3)
    with torch.no_grad():
        gnn.convs[0].lin.weight.copy_(torch.randn(d, d, generator=gg) * 0.05)
        gnn.convs[0].bias.copy_(torch.randn(d, generator=gg) * 0.05)
    convs = [(gnn.convs[0].lin.weight.detach().clone(), gnn.convs[0].bias.detach().clone())]
    model = model.to(dev).eval()
    pool = [s.tolist() for s in synth.sequences(sh, 10527, "pool", seed=2026)]
    queries = [s.tolist() for s in synth.sequences(sh, 32, "query", seed=77)]
    rng = np.random.default_rng(topk)
    idxs = [rng.permutation(10527)[:10].tolist() for _ in queries]
    ds = types.SimpleNamespace(retrieval_sources=pool)
    args = types.SimpleNamespace(fusion="graphpooling", m=1, topK=topk)
    tok = _Tok(sh.pad_id, sh.v0)
    # fused rows of the whole batch (one gather + two GEMMs) against the oracle's per-query dense GCN
    rows = generator.fusion_rows_batch(args, model, tok, ds, idxs, topk).cpu().numpy()
    for qi in (0, 13, 31):
        ref = generator_ref.fusion_graphpooling_embeds(sd, pool, queries[qi], idxs[qi], topk, convs)[0, 2].numpy()
        assert rel_err(rows[qi, 0], ref) < 1e-5 and elementwise_err(rows[qi, 0], ref) < 1
    many = generator.greedy_decode_rag_batch(args, model, tok, ds, queries, idxs, "val", 1024, 19)
    exact = 0
    for qi, (q, ix) in enumerate(zip(queries, idxs)):
        fn = lambda toks, ix=ix: generator_ref.fusion_graphpooling_embeds(sd, pool, toks, ix, topk, convs)
        want = generator_ref.greedy_decode_rag(sd, H, fn, q, sh.v0, "val")[len(q):]
        logits_at = lambda prefix, q=q, fn=fn: gpt2_ref.gpt2_forward(sd, None, H, inputs_embeds=fn(q + list(prefix)))["logits"][0, -1].numpy()
        exact += assert_tokens_equal_or_tie(many[qi][len(q):], want, logits_at, f"reddit top-{topk} query {qi}")
        assert len(many[qi]) - len(q) == len(want)
    print(f"reddit top-{topk}: {exact} of 32 generated lists identical to the oracle's")


def test_pipelined_batches_equal_one_batch_at_a_time(dev):
    """decode_rag_batches prepares batch b + 1's fusion (host half on a helper thread, device half queued before batch b's
    decode is waited for; two decoder slots): the ids must be those of the one-batch-at-a-time form, for ragged batch sizes."""
    from oracle import gpt2_ref
    from rag4dyg_amd import generator, synth
    sh = synth.SHAPES["reddit"]
    L, H, d, V = 2, 8, 512, sh.vocab_generator
    sd = gpt2_ref.make_state_dict(L, d, V, n_positions=1024, seed=56, random_affine=True)
    model = _rag_model(sd, L, H, d, V, dev)
    model.get_gnn(d, d // 2, d, 1, 0.2)
    model = model.to(dev).eval()
    pool = [s.tolist() for s in synth.sequences(sh, 600, "pool", seed=5)]
    queries = [s.tolist() for s in synth.sequences(sh, 100, "query", seed=6)]
    rng = np.random.default_rng(1)
    idxs = [rng.permutation(600)[:7].tolist() for _ in queries]
    ds = types.SimpleNamespace(retrieval_sources=pool)
    args = types.SimpleNamespace(fusion="graphpooling", m=1, topK=7)
    tok = _Tok(sh.pad_id, sh.v0)
    cuts = [0, 32, 64, 96, 100]                                        # three full batches and a ragged last one
    batches = [(queries[a:b], idxs[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    want = [generator.greedy_decode_rag_batch(args, model, tok, ds, q, ix, "val", 1024, 19) for q, ix in batches]
    got = list(generator.decode_rag_batches(args, model, tok, ds, batches, "val", 1024, 19))
    assert got == want
    torch.cuda.synchronize()
