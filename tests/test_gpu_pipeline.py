"""GPU end-to-end tests: real UCI_13 token ids against the reference's golden embeddings/scores, and the two
drop-in CLIs on a synthetic dataset written in the reference's file grammar, checked against the oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, assert_tokens_equal_or_tie, elementwise_err, load_golden, rank_mismatch_report, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _model_from_sd(sd, L, H, d, V, P, dev):
    from rag4dyg_amd.gpt2 import GPT2Config, GPT2LMHeadModelRAG
    m = GPT2LMHeadModelRAG(GPT2Config(vocab_size=V, n_positions=P, n_ctx=P, n_embd=d, n_layer=L, n_head=H))
    m.load_state_dict(sd, strict=False)
    m.tie_weights()
    return m.to(dev).eval()


def _unragged(flat, off):
    return [flat[off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]


def test_uci13_real_ids_match_reference_embeddings_scores_ranks(dev):
    """BASELINE config 2 on the real UCI_13/12 pool (1708) and test queries (110): A0-A10 end to end."""
    from oracle import gpt2_ref, retrieval_ref
    from rag4dyg_amd.retrieval import PoolIndex, encode_batches, right_pad_batches
    g, k = load_golden("g4_uci_retrieval"), load_golden("g6_uci_tokens")
    pad = int(k["pad_id"])
    sd = gpt2_ref.make_state_dict(4, 512, 1801, seed=int(g["seed"]), random_affine=True)
    m = _model_from_sd(sd, 4, 2, 512, 1801, 1024, dev)
    pool = encode_batches(m, right_pad_batches(_unragged(k["pool_flat"], k["pool_off"]), 32, pad, dev))
    q = encode_batches(m, right_pad_batches(_unragged(k["test_flat"], k["test_off"]), 32, pad, dev))
    assert rel_err(pool[:64].cpu().numpy(), g["pool_emb_head"]) < TOL
    assert rel_err(pool[-52:].cpu().numpy(), g["pool_emb_tail"]) < TOL
    assert rel_err(pool.norm(dim=1).cpu().numpy(), g["pool_emb_norms"]) < TOL
    assert rel_err(pool.double().sum(0).cpu().numpy(), g["pool_emb_colsum"]) < TOL
    assert rel_err(q.cpu().numpy(), g["query_emb"]) < TOL
    # element-wise too: |d| <= 1e-4 |ref| + 1e-5 max|ref| for EVERY element (conftest.elementwise_err)
    assert elementwise_err(q.cpu().numpy(), g["query_emb"]) < 1
    assert elementwise_err(pool[:64].cpu().numpy(), g["pool_emb_head"]) < 1
    vals, idx, S = PoolIndex(pool).search(q, 10, want_scores=True)
    assert rel_err(S.cpu().numpy(), g["scores"]) < TOL and elementwise_err(S.cpu().numpy(), g["scores"]) < 1
    # ranked top-10 against the reference's stable top-10: identical lists, except where two REFERENCE scores are closer
    # than fp32 summation noise (2e-6 on scores in [0, 1]) -- reported, and nothing beyond that is tolerated
    exact, gap = rank_mismatch_report(g["scores"], g["top10_stable"], idx.cpu().numpy())
    print(f"UCI_13 top-10: {exact:.4f} of the 110 ranked lists identical to the reference's; largest reference-score gap at a "
          f"mismatching rank {gap:.2e}; max |score - reference| {np.abs(S.cpu().numpy() - g['scores']).max():.2e}")
    assert gap <= 2e-6, gap
    assert retrieval_ref.topk_matches_modulo_ties(g["scores"], idx.cpu().numpy(), 10, 2e-6)


@pytest.mark.parametrize("split3", [True, False])
def test_trained_checkpoint_matches_reference_g10(dev, split3):
    """TRAINED weights: a checkpoint this build's trainer produced on the real UCI_13/12 data (L2 H2 d128,
    30 epochs, lr 1e-4; tools/g10_trained.py), loaded into the REFERENCE model on CPU by oracle/gen_golden.py g10 -- the 110 test
    queries and the first 256 pool histories through both: embeddings element-wise at 1e-4, scores, stable top-10 with a
    reference-score gap <= 2e-6 at any mismatching rank.  Both GEMM paths (bf16x3 split and exact-f32 MFMA).  The full-size
    (L4 H2 d512, 50 epochs) comparison is a one-off report: profiles/r03_trained_parity_full.json."""
    from rag4dyg_amd import ops
    from rag4dyg_amd.retrieval import PoolIndex, encode_batches, right_pad_batches
    g, k = load_golden("g10_trained_small"), load_golden("g6_uci_tokens")
    sd = {n[2:]: torch.from_numpy(g[n]) for n in g.files if n.startswith("w:")}
    d, V = sd["transformer.wte.weight"].shape[1], sd["transformer.wte.weight"].shape[0]
    m = _model_from_sd(sd, int(g["n_layer"]), int(g["n_head"]), d, V, sd["transformer.wpe.weight"].shape[0], dev)
    pad, NP = int(k["pad_id"]), int(g["pool_rows"])
    was = ops.gemm_split3_enabled()
    ops.set_gemm_split3(split3)
    try:
        pool = encode_batches(m, right_pad_batches(_unragged(k["pool_flat"], k["pool_off"])[:NP], 32, pad, dev))
        q = encode_batches(m, right_pad_batches(_unragged(k["test_flat"], k["test_off"]), 32, pad, dev))
        vals, idx, S = PoolIndex(pool).search(q, 10, want_scores=True)
    finally:
        ops.set_gemm_split3(was)
        m.transformer.__dict__.pop("_w3_cache", None)
    eq, ep = elementwise_err(q.cpu().numpy(), g["query_emb"]), elementwise_err(pool.cpu().numpy(), g["pool_emb"])
    exact, gap = rank_mismatch_report(g["scores"], g["top10_stable"], idx.cpu().numpy())
    print(f"G10 trained checkpoint ({'bf16x3' if split3 else 'exact f32'}): element-wise ratio queries {eq:.3f} pool {ep:.3f} (pass < 1); "
          f"max |score - reference| {np.abs(S.cpu().numpy() - g['scores']).max():.2e}; top-10 lists identical {exact:.4f}, gap {gap:.2e}")
    assert eq < 1 and ep < 1
    assert rel_err(q.cpu().numpy(), g["query_emb"]) < TOL and elementwise_err(S.cpu().numpy(), g["scores"]) < 1
    assert gap <= 2e-6, gap


# ------------------------------------------------------------------------------------------- synthetic dataset
def _write_dataset(root, ds="toy", t=4, v0=60, n_train=150, n_val=40, n_test=37, seed=0):
    rng = np.random.default_rng(seed)
    base = os.path.join(root, "resources", ds, str(t))
    os.makedirs(base)
    os.makedirs(os.path.join(root, "vocabs", ds, str(t)))
    json.dump({str(i): i for i in range(v0)}, open(os.path.join(root, "vocabs", ds, str(t), "vocab.json"), "w"))

    def hist(ego):
        parts = [f"<|endoftext|> <|history|> {ego}"]
        for s in range(int(rng.integers(1, t))):
            parts.append(f"<|time{s}|> " + " ".join(str(int(x)) for x in rng.integers(0, v0, rng.integers(0, 9))))
        return " ".join(parts).replace("  ", " ").strip() + " <|endofhistory|>"

    def pre():
        ys = rng.integers(0, 12, rng.integers(1, 4))          # small id range -> many exact Jaccard ties / positives
        return f"<|pre|> <|time{t}|> " + " ".join(str(int(y)) for y in ys) + " <|endofpre|> <|endoftext|>"
    train = [hist(int(rng.integers(0, v0))) + " " + pre() for _ in range(n_train)]
    open(os.path.join(base, "train.link_prediction"), "w").write("\n".join(train) + "\n")
    for split, n in (("val", n_val), ("test", n_test)):
        open(os.path.join(base, f"{split}.link_prediction"), "w").write(
            "\n".join(hist(int(rng.integers(0, v0))) for _ in range(n)) + "\n")
        open(os.path.join(base, f"{split}_gt.link_prediction"), "w").write("\n".join(pre() for _ in range(n)) + "\n")
    return base


def _read_matrix(path, dtype=float):
    return [list(map(dtype, ln.split())) for ln in open(path).read().splitlines() if ln.strip()]


def test_annotation_cli_end_to_end_against_oracle(dev, tmp_path, monkeypatch):
    from oracle import jaccard_ref
    from rag4dyg_amd import annotation
    base = _write_dataset(str(tmp_path))
    monkeypatch.chdir(tmp_path)
    np.random.seed(0)
    annotation.main(["retrieval_data_annotation.py", "toy", "4", "0.8"])
    rd = jaccard_ref.read_lines
    train = rd(os.path.join(base, "train.link_prediction"))
    tr_in, tr_out = jaccard_ref.get_inout_list(train, train)
    out_dir = tmp_path / "resources" / "toy" / "4" / "train_retrieval"
    for split in ("test", "val"):
        _, s_out = jaccard_ref.get_inout_list(rd(os.path.join(base, f"{split}.link_prediction")),
                                              rd(os.path.join(base, f"{split}_gt.link_prediction")))
        ref = jaccard_ref.occurrence_matrix(s_out, tr_out)
        got_scores = np.array(_read_matrix(out_dir / f"{split}_score.retrieval"))
        assert np.array_equal(got_scores, ref)                          # str(float64) round-trips exactly
        got_idx = np.array(_read_matrix(out_dir / f"{split}_index.retrieval", int))
        assert np.array_equal(got_idx, jaccard_ref.rank_rows(ref))       # full permutation, canonical order
        # text identical to the reference writer's (retrieval_data_annotation.py:92-93)
        first = open(out_dir / f"{split}_score.retrieval").readline().rstrip("\n")
        assert first == ' '.join(str(x) for x in ref[0])
    m_out = jaccard_ref.occurrence_matrix(tr_out, tr_out); np.fill_diagonal(m_out, 0)
    m_in = jaccard_ref.occurrence_matrix(tr_in, tr_in); np.fill_diagonal(m_in, 0)
    gen_dir = tmp_path / "resources" / "train_generator" / "toy" / "4" / "train_gt_topk"
    top = np.array(_read_matrix(gen_dir / "train_index.gen", int))
    assert np.array_equal(top, jaccard_ref.rank_rows(m_out)[:, :10])
    assert np.array_equal(np.array(_read_matrix(gen_dir / "train_score.gen")),
                          np.take_along_axis(m_out, jaccard_ref.rank_rows(m_out)[:, :10], axis=1))
    pos = jaccard_ref.train_positives(m_out, 0.8)
    cands = jaccard_ref.train_negative_candidates(m_out, m_in, 0.8)
    triples = _read_matrix(out_dir / "train_index.retrieval", int)
    assert len(triples) == sum(len(p) for p in pos) > 0
    seen = {}
    for i, p, n in triples:
        seen.setdefault(i, []).append(p)
        assert n in cands[i]
    assert all(seen.get(i, []) == p for i, p in enumerate(pos))
    sc = _read_matrix(out_dir / "train_score.retrieval")
    assert all(s[1] == m_out[int(t_[0]), t_[1]] and s[2] == m_out[int(t_[0]), t_[2]] for s, t_ in zip(sc, triples))


@pytest.mark.parametrize("world", [2, 3])
def test_annotation_cli_rows_sharded_over_ranks_equal_one_process_byte_for_byte(dev, tmp_path, world):
    """the target rows of the four Jaccard matrices are dealt to the ranks (one process per GPU; all on
    cuda:0 here), the source CSR is replicated, rank 0 concatenates the row-range files: every output file equals the
    one-process run byte for byte (random negative picks included: they are seeded per anchor row)."""
    import hashlib
    import shutil
    one, many = tmp_path / "one", tmp_path / "many"
    _write_dataset(str(one), n_train=157, n_val=41, n_test=37, seed=11)
    shutil.copytree(one, many)
    cli = os.path.join(REPO, "retrieval_data_annotation.py")
    code = f"import sys, numpy as np; sys.path.insert(0, {REPO!r}); np.random.seed(5); sys.argv = ['x', 'toy', '4', '0.5']; " \
           f"from rag4dyg_amd.annotation import main; main(sys.argv)"
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, "-c", code], cwd=one, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-1500:]
    procs = [subprocess.Popen([sys.executable, "-c", code], cwd=many, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                                       MASTER_PORT=str(29560 + world))) for r in range(world)]
    done = [pr.communicate(timeout=600) for pr in procs]
    assert all(pr.returncode == 0 for pr in procs), [e[-1500:] for _, e in done]
    assert "Number of positive samples" in done[0][0] and p.stdout.splitlines()[-3:] == done[0][0].splitlines()[-3:]
    names = []
    for root, _dirs, fs in os.walk(one / "resources"):
        for f in fs:
            if f.endswith((".retrieval", ".gen")):
                names.append(os.path.relpath(os.path.join(root, f), one))
    assert len(names) == 8, names
    for rel in names:
        a, b = (one / rel).read_bytes(), (many / rel).read_bytes()
        assert len(a) > 0 and hashlib.sha256(a).digest() == hashlib.sha256(b).digest(), rel
    assert not [f for _r, _d, fs in os.walk(many) for f in fs if ".part" in f]          # parts removed after the join
    assert os.path.exists(cli)


def test_main_retriever_cli_end_to_end_against_oracle(dev, tmp_path, monkeypatch):
    import main_retriever
    from oracle import gpt2_ref, jaccard_ref, retrieval_ref
    from rag4dyg_amd.tokenizer import build_tokenizer
    base = _write_dataset(str(tmp_path), seed=3)
    monkeypatch.chdir(tmp_path)
    rd = jaccard_ref.read_lines
    train = rd(os.path.join(base, "train.link_prediction"))
    _, tr_out = jaccard_ref.get_inout_list(train, train)
    gt = {}
    for split in ("test", "val"):
        _, s_out = jaccard_ref.get_inout_list(rd(os.path.join(base, f"{split}.link_prediction")),
                                              rd(os.path.join(base, f"{split}_gt.link_prediction")))
        gt[split] = jaccard_ref.occurrence_matrix(s_out, tr_out)
        open(os.path.join(base, f"{split}_score.retrieval"), "w").write(
            "\n".join(' '.join(str(x) for x in row) for row in gt[split]) + "\n")
    L, H, d = 2, 2, 64                               # len(tok) = V0 + 8 + t = 60 + 12
    tok, _ = build_tokenizer("toy", 4)
    assert len(tok) == 72
    sd = gpt2_ref.make_state_dict(L, d, len(tok), seed=77, random_affine=True)
    ck = tmp_path / "out" / "checkpoint-0"
    ck.mkdir(parents=True)
    full_sd = dict(sd)
    for i in range(L):                               # reference checkpoints carry the causal buffers (strict load)
        full_sd[f"transformer.h.{i}.attn.bias"] = torch.tril(torch.ones(1024, 1024)).view(1, 1, 1024, 1024)
    torch.save(full_sd, ck / "pytorch_model.bin")
    argv = (f"--dataset toy --timestamp 4 --output_dir {tmp_path}/out --model_type gpt2 --model_name_or_path gpt2 "
            f"--train_data_file {base}/train.link_prediction --do_eval --eval_all_checkpoints "
            f"--eval_data_file {base}/val.link_prediction --eval_data_gt_file {base}/val_score.retrieval "
            f"--test_data_file {base}/test.link_prediction --test_data_gt_file {base}/test_score.retrieval "
            f"--block_size 512 --n_layer {L} --n_head {H} --n_embed {d} --topK 5").split()
    captured = {}
    real_test = main_retriever.test

    def spy(epoch, args, model, tokenizer, evaluate=False, prefix=""):      # also run the validation branch (A11: BCE metric)
        captured["val"] = real_test(epoch, args, model, tokenizer, evaluate=True, prefix="probe")
        return real_test(epoch, args, model, tokenizer, evaluate=evaluate, prefix=prefix)
    monkeypatch.setattr(main_retriever, "test", spy)
    main_retriever.main(argv)
    monkeypatch.setattr(main_retriever, "test", real_test)
    res = tmp_path / "resources" / "retrieval_result" / "toy"
    idx_rows = np.array(_read_matrix(res / "test_index.gen", int))
    score_rows = np.array(_read_matrix(res / "test_score.gen"))
    # binary side-cars: same rows as the text (scores to the text's 4 decimals); stale side-car is ignored
    from rag4dyg_amd.retriever import read_matrix_rows
    assert (res / "test_index.gen.bin").exists() and (res / "test_score.gen.bin.json").exists()
    assert np.array_equal(np.array(read_matrix_rows(str(res / "test_index.gen"), int)), idx_rows)
    assert np.abs(np.array(read_matrix_rows(str(res / "test_score.gen"), float)) - score_rows).max() <= 5.1e-5
    with open(res / "test_index.gen.bin", "ab") as fh:
        fh.write(b"\0\0\0\0")                                        # wrong size -> falls back to the text file
    assert np.array_equal(np.array(read_matrix_rows(str(res / "test_index.gen"), int)), idx_rows)
    # oracle pipeline on the same text
    pool_ids = tok([ln.split('<|pre|>')[0].strip() for ln in train], max_length=512)["input_ids"]
    q_ids = tok(rd(os.path.join(base, "test.link_prediction")), max_length=512)["input_ids"]
    pool = retrieval_ref.encode_batches(sd, H, retrieval_ref.right_pad_batches(pool_ids, 32, tok.pad_token_id))
    q = retrieval_ref.encode_batches(sd, H, retrieval_ref.right_pad_batches(q_ids, 32, tok.pad_token_id))
    S = retrieval_ref.score_batch(q, pool).numpy()
    assert score_rows.shape == S.shape == (37, 150)
    # A11 (train_retriever.py:439-441,477): BCEWithLogits of the score rows against the float32 Jaccard rows, summed over
    # the batches' means and divided by the number of queries -- against oracle/retrieval_ref.bce_with_logits_mean
    qv_ids = tok(rd(os.path.join(base, "val.link_prediction")), max_length=512)["input_ids"]
    qv = retrieval_ref.encode_batches(sd, H, retrieval_ref.right_pad_batches(qv_ids, 32, tok.pad_token_id))
    Sv = retrieval_ref.score_batch(qv, pool).numpy()
    ref_loss = sum(retrieval_ref.bce_with_logits_mean(Sv[i:i + 32], gt["val"][i:i + 32]) for i in range(0, len(Sv), 32)) / len(Sv)
    val_metrics, val_loss = captured["val"]
    assert abs(float(val_loss) - ref_loss) <= 1e-5 * ref_loss, (float(val_loss), ref_loss)
    ref_val_hits = retrieval_ref.hit_metrics([Sv[i:i + 32] for i in range(0, len(Sv), 32)],
                                             [gt["val"][i:i + 32] for i in range(0, len(Sv), 32)])
    assert abs(val_metrics["hit@1"] - ref_val_hits[0]) <= 1e-4 and abs(val_metrics["hit@3"] - ref_val_hits[1]) <= 1e-4
    assert np.abs(score_rows - S).max() < 1e-4 + 5e-5                   # %.4f text
    assert all(sorted(r) == list(range(150)) for r in idx_rows.tolist())   # full permutations
    exact5, gap5 = rank_mismatch_report(S, retrieval_ref.rank_full(S)[:, :5], idx_rows[:, :5])
    print(f"CLI top-5: {exact5:.4f} of rows identical to the oracle's, largest oracle-score gap at a mismatch {gap5:.2e}")
    assert gap5 <= 2e-6, gap5
    csv = open(res / "test_results.csv").read()
    assert "Hit@1, Hit@3" in csv
    hits = [float(x) for x in csv.strip().splitlines()[1].split(",")[-2:]]
    sb = [S[i:i + 32] for i in range(0, 37, 32)]
    gb = [gt["test"][i:i + 32] for i in range(0, 37, 32)]
    ref_hits = retrieval_ref.hit_metrics(sb, gb)
    # hit@k: equal to the oracle's unless a rank mismatch above (none beyond 2e-6 ties) flips a hit -- at most one query per
    # flipped row, i.e. 1/37 per batch-mean term; with identical rankings the metrics are identical
    tol_hit = 0.0 if exact5 == 1.0 else (1.0 - exact5) + 1e-4
    assert abs(hits[0] - ref_hits[0]) <= tol_hit + 1e-4 and abs(hits[1] - ref_hits[1]) <= tol_hit + 1e-4, (hits, ref_hits, exact5)
    # tokenizer files written in the reference layout
    assert (tmp_path / "tokenizers" / "toy" / "4" / "tokenizer.json").exists()
    # one process per GPU (two ranks, gloo on this one-GPU box): the pool encode is sharded by whole batches, every rank
    # scores all queries, rank 0 writes -- same rankings and metrics as the single process
    os.remove(res / "test_index.gen"); os.remove(res / "test_results.csv")
    procs = []
    for rk in range(2):
        env = dict(os.environ, R4D_DIST_BACKEND="gloo", PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""),
                   RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29534")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "main_retriever.py")] + argv, cwd=tmp_path, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    done = [pr.communicate(timeout=600) for pr in procs]
    assert all(pr.returncode == 0 for pr in procs), [e[-1500:] for _, e in done]
    idx2 = np.array(_read_matrix(res / "test_index.gen", int))
    assert idx2.shape == idx_rows.shape
    assert rank_mismatch_report(S, retrieval_ref.rank_full(S)[:, :5], idx2[:, :5])[1] <= 2e-6
    assert np.abs(np.array(_read_matrix(res / "test_score.gen")) - score_rows).max() <= 1.01e-4
    assert open(res / "test_results.csv").read() == csv


_RANK_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import rag4dyg_amd.dist as rdist
from rag4dyg_amd import ops
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=rank, world_size=world)
dev = torch.device("cuda:0")
_orig = rdist.all_gather_cat
rdist.all_gather_cat = lambda t, group=None: _orig(t.cpu(), group).to(t.device)   # collectives staged through host (1-GPU box)
g = torch.Generator().manual_seed(0)
Q, N, d, k = 64, 5000, 256, 10
q = ops.normalize_rows(torch.randn(Q, d, generator=g).to(dev))
p = torch.randn(N, d, generator=g); p[4100] = p[77]
p = ops.normalize_rows(p.to(dev))
s, e = rdist.shard_bounds(N, world)[rank]
q_local = q[rank * (Q // world):(rank + 1) * (Q // world)].contiguous()
q_all = rdist.all_gather_cat(q_local)                      # embeddings all-gather
assert torch.equal(q_all, q)
vals, idx = rdist.sharded_topk(q_all, p[s:e].contiguous(), s, k,
                               lambda a, b, kk, off: ops.score_topk(a, b, kk, off)[:2], ops.merge_topk)
rv, ri, _ = ops.score_topk(q, p, k)
assert torch.equal(idx, ri) and torch.equal(vals, rv), rank
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 4])
def test_pool_sharded_over_ranks_equals_single_gpu(dev, tmp_path, world):
    """(e) one process per rank (all on cuda:0 here, gloo rendezvous): HIP local scan + all-gather + HIP merge
    reproduces the single-GPU top-k bit for bit, ties included."""
    import subprocess, sys
    from conftest import REPO
    script = tmp_path / "rank_worker.py"
    script.write_text(_RANK_WORKER)
    port = str(29700 + world + os.getpid() % 100)
    procs = [subprocess.Popen([sys.executable, str(script), REPO, port],
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(world)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


def test_simpledyg_greedy_eval_matches_oracle_decode(dev, tmp_path, monkeypatch):
    """greedy link-prediction eval -- generated token ids equal the oracle's CPU decode, and the CLI
    writes the reference's result files."""
    import main_SimpleDyG
    from oracle import gpt2_ref, jaccard_ref
    from rag4dyg_amd.evaluation import Evaluation, greedy_decode
    from rag4dyg_amd.gpt2 import GPT2Config, GPT2LMHeadModel
    from rag4dyg_amd.tokenizer import build_tokenizer
    base = _write_dataset(str(tmp_path), n_train=20, n_val=12, n_test=12, seed=5)
    monkeypatch.chdir(tmp_path)
    tok, _ = build_tokenizer("toy", 4, with_mask=False)
    L, H, d, P = 2, 2, 64, 128
    sd = gpt2_ref.make_state_dict(L, d, len(tok), n_positions=P, seed=9, random_affine=True)
    cfg = GPT2Config(vocab_size=len(tok), n_positions=P, n_ctx=P, n_embd=d, n_layer=L, n_head=H)
    m = GPT2LMHeadModel(cfg)
    m.load_state_dict(sd, strict=False); m.tie_weights()
    ck = tmp_path / "out" / "checkpoint-0"
    ck.mkdir(parents=True)
    m.save_pretrained(str(ck))
    m = m.to(dev).eval()
    lines = jaccard_ref.read_lines(os.path.join(base, "val.link_prediction"))
    eos = tok.eos_token_id
    def logits_for(ids):
        return lambda prefix: gpt2_ref.gpt2_forward(sd, torch.tensor([ids + list(prefix)]), H)["logits"][0, -1].numpy()
    for mode in ("val", "test"):
        for ln in lines[:6]:
            ids = tok.encode(ln)
            got = greedy_decode(m, tok, ids, mode, P, 12, dev)
            ref = gpt2_ref.greedy_decode(sd, H, ids, eos, mode, P, 12)
            assert got[:len(ids)] == ids and len(got) > len(ids)
            assert_tokens_equal_or_tie(got[len(ids):], ref[len(ids):], logits_for(ids), f"SimpleDyG {mode}")   # equal, or a printed sub-2e-6 tie
    from rag4dyg_amd.evaluation import greedy_decode_batch
    prompts = [tok.encode(ln) for ln in lines[:9]]
    for mode in ("val", "test"):                        # cached, batched decode == the oracle's one-at-a-time loop
        many = greedy_decode_batch(m, tok, prompts, mode, P, 12, dev)
        for p_, got in zip(prompts, many):
            ref = gpt2_ref.greedy_decode(sd, H, p_, eos, mode, P, 12)
            assert_tokens_equal_or_tie(got[len(p_):], ref[len(p_):], logits_for(p_), f"SimpleDyG {mode} cached batch")
    argv = (f"--dataset toy --timestamp 4 --output_dir {tmp_path}/out --model_type gpt2 --train_data_file "
            f"{base}/train.link_prediction --do_eval --eval_all_checkpoints --eval_data_file {base}/val.link_prediction "
            f"--eval_data_gt_file {base}/val_gt.link_prediction --block_size 128 --n_layer {L} --n_head {H} --n_embed {d}").split()
    res = main_SimpleDyG.main(argv)
    r = next(iter(res.values()))
    assert 0.0 <= r["NDCG"][0] <= 1.0 and 0.0 <= r["jaccard"][0] <= 1.0 and r["eval_loss"] > 0
    out_dir = tmp_path / "out" / "results" / "test_score"
    assert (out_dir / "test_results_epoch.csv").exists() and (out_dir / "eval_results_0.json").exists()
    hdr = open(out_dir / "test_results_epoch.csv").readline()
    assert hdr.startswith("dataset,method,time,nlayer,nhead,nemb,bz,lr,seed,NDCG@5,jaccard@5")
    # one process per GPU (two ranks, gloo on this one-GPU box): every other batch of 4 per rank, same predictions
    single = json.load(open(out_dir / "eval_results_0.json"))
    procs = []
    for rk in range(2):
        env = dict(os.environ, R4D_DIST_BACKEND="gloo", PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""),
                   RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29535")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "main_SimpleDyG.py")] + argv +
                                      ["--per_gpu_eval_batch_size", "4"], cwd=tmp_path, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    done = [pr.communicate(timeout=600) for pr in procs]
    assert all(pr.returncode == 0 for pr in procs), [e[-1500:] for _, e in done]
    double = json.load(open(out_dir / "eval_results_0.json"))
    assert double.keys() == single.keys()
    assert sum(double[k].get("predicted") == single[k].get("predicted") for k in single) >= len(single) - 1


def test_rag_generator_fusion_and_decode_match_oracle(dev, tmp_path, monkeypatch):
    """graph-pooling and MLP fusion rows within 1e-5 of the oracle, greedy RAG decode ids equal to the
    oracle's CPU decode, and the main_generator CLI evaluates a checkpoint and writes the reference's files."""
    import types
    import main_generator
    from oracle import generator_ref, gpt2_ref, jaccard_ref
    from rag4dyg_amd import generator
    from rag4dyg_amd.gpt2 import GPT2Config, GPT2LMHeadModelRAG
    from rag4dyg_amd.tokenizer import build_tokenizer
    base = _write_dataset(str(tmp_path), n_train=40, n_val=12, n_test=12, seed=7)
    monkeypatch.chdir(tmp_path)
    tok, _ = build_tokenizer("toy", 4, with_mask=False)
    L, H, d, P, m_rows, topk = 2, 2, 64, 160, 3, 5
    sd = gpt2_ref.make_state_dict(L, d, len(tok), n_positions=P, seed=11, random_affine=True)
    cfg = GPT2Config(vocab_size=len(tok), n_positions=P, n_ctx=P, n_embd=d, n_layer=L, n_head=H)
    model = GPT2LMHeadModelRAG(cfg)
    model.load_state_dict(sd, strict=False); model.tie_weights()
    g = torch.Generator().manual_seed(5)
    gnn = model.get_gnn(d, d // 2, d, 1, 0.2)
    mlp = model.get_mlp(512, m_rows, 2)
    with torch.no_grad():
        gnn.convs[0].lin.weight.copy_(torch.randn(d, d, generator=g) * 0.2)
        gnn.convs[0].bias.copy_(torch.randn(d, generator=g) * 0.1)
        for mod in mlp.layers:
            if hasattr(mod, "weight"):
                mod.weight.copy_(torch.randn(mod.weight.shape, generator=g) * 0.05)
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
    convs = [(gnn.convs[0].lin.weight.detach().clone(), gnn.convs[0].bias.detach().clone())]
    layers = [(mod.weight.detach().clone(), mod.bias.detach().clone()) for mod in mlp.layers if hasattr(mod, "weight")]
    ck = tmp_path / "gen" / "checkpoint-0"
    ck.mkdir(parents=True)
    model.mlp_fusion = None                             # a graph-pooling generator checkpoint holds gnn_fusion.* only
    model.save_pretrained(str(ck))
    model.mlp_fusion = mlp
    model = model.to(dev).eval()
    train_lines = jaccard_ref.read_lines(os.path.join(base, "train.link_prediction"))
    sources = tok(train_lines, add_special_tokens=True, max_length=128)["input_ids"]
    ds = types.SimpleNamespace(retrieval_sources=sources)
    lines = jaccard_ref.read_lines(os.path.join(base, "val.link_prediction"))
    rng = np.random.default_rng(1)
    for fusion in ("graphpooling", "mlp"):
        args = types.SimpleNamespace(fusion=fusion, m=m_rows, topK=topk)
        for ln in lines[:4]:
            ids = tok.encode(ln)
            idx = rng.permutation(len(sources))[:9].tolist()
            if fusion == "graphpooling":
                ref_aug = generator_ref.fusion_graphpooling_embeds(sd, sources, ids, idx, topk, convs)
                fn = lambda t, idx=idx: generator_ref.fusion_graphpooling_embeds(sd, sources, t, idx, topk, convs)
                nrows = 1
            else:
                ref_aug = generator_ref.fusion_mlp_embeds(sd, sources, ids, idx, topk, m_rows, layers, tok.pad_token_id)
                fn = lambda t, idx=idx: generator_ref.fusion_mlp_embeds(sd, sources, t, idx, topk, m_rows, layers, tok.pad_token_id)
                nrows = m_rows
            rows = generator.fusion_rows(args, model, tok, ds, idx, topk).cpu()
            assert rows.shape == (nrows, d)
            assert rel_err(rows.numpy(), ref_aug[0, 2:2 + nrows].numpy()) < 1e-5
            got = generator.greedy_decode_rag(args, model, tok, ds, ids, idx, "val", P, 12)
            ref = generator_ref.greedy_decode_rag(sd, H, fn, ids, tok.eos_token_id, "val", P, 12)
            assert got[:len(ids)] == ids and len(got) > len(ids)
            la = lambda prefix, fn=fn, ids=ids: gpt2_ref.gpt2_forward(sd, None, H, inputs_embeds=fn(ids + list(prefix)))["logits"][0, -1].numpy()
            assert_tokens_equal_or_tie(got[len(ids):], ref[len(ids):], la, f"RAG {fusion}")
    # data-parallel decode of many queries (right-padded batch) == one query at a time, val and test stop rules
    args = types.SimpleNamespace(fusion="graphpooling", m=1, topK=topk)
    qs = [tok.encode(ln) for ln in lines[:9]]
    ixs = [rng.permutation(len(sources))[:9].tolist() for _ in qs]
    rows_one = torch.stack([generator.fusion_rows(args, model, tok, ds, ix, topk) for ix in ixs])
    rows_all = generator.fusion_rows_batch(args, model, tok, ds, ixs, topk)
    assert rows_all.shape == rows_one.shape and rel_err(rows_all.cpu().numpy(), rows_one.cpu().numpy()) < 1e-5
    for mode in ("val", "test"):
        many = generator.greedy_decode_rag_batch(args, model, tok, ds, qs, ixs, mode, P, 12)
        for q, ix, got in zip(qs, ixs, many):
            fn = lambda t, ix=ix: generator_ref.fusion_graphpooling_embeds(sd, sources, t, ix, topk, convs)
            ref = generator_ref.greedy_decode_rag(sd, H, fn, q, tok.eos_token_id, mode, P, 12)
            la = lambda prefix, fn=fn, q=q: gpt2_ref.gpt2_forward(sd, None, H, inputs_embeds=fn(q + list(prefix)))["logits"][0, -1].numpy()
            assert_tokens_equal_or_tie(got[len(q):], ref[len(q):], la, f"RAG cached batch {mode}")
        if mode == "test":
            assert all(len(t) <= P - 12 for t in many) and any(len(t) == P - 12 or t[-1] == tok.eos_token_id for t in many)
    # CLI: index / score files as main_retriever writes them (one row of pool indices / scores per query)
    n_q = len(jaccard_ref.read_lines(os.path.join(base, "test.link_prediction")))
    os.makedirs("resources/retrieval_result/toy", exist_ok=True)
    with open("resources/retrieval_result/toy/test_index.gen", "w") as fi, open("resources/retrieval_result/toy/test_score.gen", "w") as fs:
        for _ in range(n_q):
            perm = rng.permutation(len(sources))
            fi.write(" ".join(str(int(v)) for v in perm) + "\n")
            fs.write(" ".join(f"{v:.4f}" for v in rng.random(len(sources))) + "\n")
    os.makedirs("resources/train_generator/toy/4/train_gt_topk", exist_ok=True)      # read by TextIndexScoreDataset
    with open("resources/train_generator/toy/4/train_gt_topk/train_index.gen", "w") as fi, \
            open("resources/train_generator/toy/4/train_gt_topk/train_score.gen", "w") as fs:
        for _ in range(len(sources)):
            fi.write(" ".join(str(int(v)) for v in rng.permutation(len(sources))[:10]) + "\n")
            fs.write(" ".join(f"{v:.4f}" for v in rng.random(10)) + "\n")
    argv = (f"--dataset toy --timestamp 4 --output_dir {tmp_path}/gen --model_type gpt2 --fusion graphpooling --topK {topk} "
            f"--train_index_file resources/train_generator/toy/4/train_gt_topk/train_index.gen "
            f"--train_score_file resources/train_generator/toy/4/train_gt_topk/train_score.gen "
            f"--gnn_layers 1 --m 1 --train_data_file {base}/train.link_prediction --do_eval --eval_all_checkpoints "
            f"--eval_data_file {base}/val.link_prediction --eval_data_gt_file {base}/val_gt.link_prediction "
            f"--test_data_file {base}/test.link_prediction --test_data_gt_file {base}/test_gt.link_prediction "
            f"--test_index_file resources/retrieval_result/toy/test_index.gen "
            f"--test_score_file resources/retrieval_result/toy/test_score.gen "
            f"--block_size 128 --n_layer {L} --n_head {H} --n_embed {d} --config_name {ck}").split()
    res = main_generator.main(argv)
    r = next(iter(res.values()))
    assert all(0.0 <= r[k][0] <= 1.0 for k in ("R", "NDCG", "jaccard"))
    outs = list((tmp_path / "rag_results" / "val_mode" / "toy" / "4").glob("*/results/test_score/eval_results.json"))
    assert len(outs) == 1
    # data-parallel over the test queries: two processes (torch.distributed.run, gloo here: one GPU) decode every other
    # batch of 4 and write the same predictions and metrics as the single-process run
    single = json.load(open(outs[0]))
    procs = []
    for rk in range(2):           # what torch.distributed.run sets per rank (its own parser trips over the reference's `--m`)
        env = dict(os.environ, R4D_DIST_BACKEND="gloo", PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""),
                   RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "main_generator.py")] + argv +
                                      ["--per_gpu_eval_batch_size", "4"], cwd=tmp_path, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    done = [pr.communicate(timeout=600) for pr in procs]
    assert all(pr.returncode == 0 for pr in procs), [e[-1500:] for _, e in done]
    p = types.SimpleNamespace(stdout=done[0][0])
    double = json.load(open(outs[0]))
    assert double.keys() == single.keys() and len(single) == n_q
    assert sum(double[k].get("predicted") == single[k].get("predicted") for k in single) >= len(single) - 1
    assert f"'R': [{r['R'][0]}]" in p.stdout or sum(double[k] == single[k] for k in single) < len(single)


def test_bench_contract_one_rank_and_two_rank_rehearsal(tmp_path):
    """bench.py prints ONE JSON line with the contract's fields (roofline and cpu_baseline included) at N = 1, and its N > 1
    control flow -- pool shard per rank, pipelined collectives drained inside the timed region, MAX over ranks -- runs with
    two ranks on this one GPU (gloo rehearsal backend; the self-test inside compares pipelined with synchronous results)."""
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    small = ["--steps", "3", "--warmup", "1", "--pool-per-gpu", "2048", "--query-batches", "8"]
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + small, env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and d["vs_baseline"] is None
    assert d["dtype"].startswith("f32 (bf16x3") and d["extras"]["gemm"] == "split3"
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] <= 1 and d["cpu_baseline"]["kind"] == "port"
    assert d["roofline"]["kernel"].startswith("gemm_s3") and d["roofline"]["peak"] == 416.7     # bf16 dense peak / 6 products
    ex = d["extras"]["exact_f32"]                                        # the same timed region on the exact-f32 MFMA kernels
    assert ex["dtype"] == "f32" and ex["value"] > 0 and ex["roofline"]["peak"] == 157.3 and 0 < ex["roofline"]["frac"] <= 1
    assert ex["top10_rows_identical_to_split3"] >= 0.99 and ex["max_abs_score_diff_vs_split3"] < 2e-6
    v1 = d["extras"]["verify"]                                           # N = 1: last timed step against the oracle, in the run
    assert v1.get("error") is None and v1["pass"] is True and v1["queries"] == 256 and v1["rows_identical_to_oracle"] >= 0.98, v1
    assert v1["max_oracle_score_gap_at_mismatch"] <= 2e-6 and v1["timed_step_equals_recomputation"] is True
    assert v1["scan_path_rows_identical_to_gemm_path"] >= 0.98 and v1["scan_vs_gemm_max_abs_score_diff"] < 1e-6
    assert "lm_head" in d["cpu_baseline"]["sample"] and d["cpu_baseline"]["value_without_lm_head"] >= d["cpu_baseline"]["value"]
    sq = d["extras"]["scan_q32"]                                         # the north-star kernel on its own (HBM) roofline
    assert sq["queries"] == 32 and sq["pool_rows"] == 2048 and sq["roofline"]["bound"] == "hbm" and 0 < sq["roofline"]["frac"] <= 1
    assert sq["roofline"]["frac"] == sq["roofline"]["frac_survey"] <= sq["roofline"]["frac_moved"]      # SURVEY's B_score vs bytes moved
    assert sq["algorithmic_bytes"] == 4 * 2048 * 512 + 4 * 32 * 512 + 8 * 32 * 10 and 0 < sq["roofline"]["frac_scan_plus_topk"] < sq["roofline"]["frac"]
    assert d["roofline"]["traffic"] is None                              # PMC figure belongs to another workload / pool size
    assert len(d["extras"]["source_sha"]) == 16 and d["config"]["pool_rows_total"] == 2048
    lb = d["extras"]["length_bucketed"]                                  # SURVEY 8d's second run, never the headline
    assert lb["value"] > 0 and lb["mean_padded_T"] > 0 and "not parity-comparable" in lb["note"]
    env2 = dict(env, R4D_BENCH_BACKEND="gloo", R4D_BENCH_PIPELINE="force")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", "29536", os.path.join(REPO, "bench.py"), "--gpus", "2",
                        "--no-cpu-baseline"] + small, env=env2, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                              # rank 0 only
    d2 = json.loads(lines[0])
    assert d2["n_gpus"] == 2 and d2["scaling"] == "weak" and d2["value"] > 0 and d2["config"]["pool_rows_total"] == 4096
    v = d2["extras"]["verify"]                                           # post-run check: sharded == one-GPU recomputation
    assert v.get("error") is None and v["world_size"] == 2 and v["sharded_topk_equals_one_gpu"] is True, v
    assert v["pool_rows_checked"] == 4096 and len(v["per_rank_ms_per_step"]) == 2
    assert v["collectives"].get("error") is None and v["collectives"]["all_gather_embeddings_us"] > 0, v["collectives"]
    assert "pipeline" in d2["config"]["collectives"] and "unavailable" not in p.stderr
