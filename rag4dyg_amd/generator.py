"""RAG generator INFERENCE: fusion of the retrieved demonstrations + greedy link prediction.

Host-side mirror of ``dataloader/generator.py`` (``TextIndexScoreDataset``), ``utils/model.py:105-224``
(``fusion_mlp`` / ``fusion_graphpooling``) and ``utils/Evaluation_generator.py:49-265``
(``get_eval_metrics_generator``), consuming the ``*_index.gen`` / ``*_score.gen`` files the retrieve path of this build
writes.  All dense work runs on the gfx950 kernels: the GCN / MLP projections and the normalised-adjacency product
through ``r4d_conv1d_f32``, the forward on ``inputs_embeds`` through ``r4d_gpt2_encode_f32``, the tied lm_head on the
LAST row only through ``r4d_lm_logits_f32``.  torch is used for the embedding gathers and the concatenation (plumbing).

Differences from the reference, all value-preserving:
  * the fused rows depend only on the retrieved indices, so they are computed ONCE per query instead of once per
    generated token (``Evaluation_generator.py:155`` re-runs the fusion every step);
  * the GCN runs on a dense normalised adjacency (the fused graph of top-K sequences has a few hundred nodes) instead
    of torch_geometric's scatter -- same sums, ``D^-1/2 (A+I) D^-1/2`` as GCNConv's defaults define it;
  * GNNs with more than one layer: the reference applies ``F.dropout`` with ``training=True`` even at inference
    (``modeling_rag.py:69``, stochastic); this build applies none (the expectation).  Shipped scripts use one layer.
Training (``train/train_generator.py``) is out of scope: ``main_generator.py --do_train`` raises.
"""
import json
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .dataloader import read_nonblank_lines
from .evaluation import Evaluation


# ------------------------------------------------------------------------------------------------ fusion modules
class GCNConvParams(nn.Module):
    """Parameter holder with torch_geometric's GCNConv names: ``lin.weight`` [out, in] (PyG >= 2.0) and ``bias``."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.lin = nn.Linear(in_dim, out_dim, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_dim))
        nn.init.xavier_uniform_(self.lin.weight)           # glorot, as GCNConv.reset_parameters

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        old = prefix + "weight"                           # PyG 1.7.x: ``weight`` [in, out]
        if old in state_dict and prefix + "lin.weight" not in state_dict:
            state_dict[prefix + "lin.weight"] = state_dict.pop(old).t().contiguous()
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class GNN(nn.Module):
    """``models/modeling_rag.py:44-71`` (layer widths; the forward lives in ``gnn_pool``)."""

    def __init__(self, input_dim, hidden_dim, output_dim, n_layers, dropout_rate=0.5):
        super().__init__()
        self.n_layers, self.dropout_rate = n_layers, dropout_rate
        dims = [input_dim] + [hidden_dim] * (n_layers - 1) + [output_dim]
        self.convs = nn.ModuleList(GCNConvParams(dims[i], dims[i + 1]) for i in range(n_layers))


class MLP_custom(nn.Module):
    """``models/modeling_rag.py:74-99``: same ``layers`` Sequential indices, so reference state_dicts load."""

    def __init__(self, input_dim, output_dim, n_layers):
        super().__init__()
        self.n_layers = n_layers
        hidden = int(input_dim / 2)
        if n_layers == 1:
            self.layers = nn.Sequential(nn.Linear(input_dim, output_dim))
        else:
            mods = [nn.Linear(input_dim, hidden), nn.ReLU()]
            for _ in range(n_layers - 2):
                mods += [nn.Linear(hidden, hidden), nn.ReLU()]
            mods.append(nn.Linear(hidden, output_dim))
            self.layers = nn.Sequential(*mods)


def _linear(x, lin, epilogue="none"):
    """y = x W^T + b on the k-contiguous GEMM (``W`` [out, in] IS the transposed Conv1D weight)."""
    w = lin.weight
    b = lin.bias if lin.bias is not None else torch.zeros(w.shape[0], device=w.device)
    return ops.conv1d(x.contiguous(), w.t().contiguous(), b, epilogue, None, w.contiguous())


# ------------------------------------------------------------------------------------------------ graph fusion
def star_union_graph(retrieval_sources, idxs):
    """Fused graph of ``fusion_graphpooling`` (``utils/model.py:181-189``): ego (token 2 of each retrieved sequence)
    linked to every token id of that sequence.  Node order = networkx insertion order.  Returns (nodes, edge set)."""
    order, pos, edges = [], {}, set()
    for n in idxs:
        seq = [int(e) for e in retrieval_sources[int(n)]]
        ego = seq[2]
        for v in (ego,):
            if v not in pos:
                pos[v] = len(order); order.append(v)
        a = pos[ego]
        for e in seq:
            if e not in pos:
                pos[e] = len(order); order.append(e)
            b = pos[e]
            edges.add((a, b) if a <= b else (b, a))
    return order, edges


def gcn_norm_dense(n, edges, device):
    """``D^-1/2 (A + I) D^-1/2`` (GCNConv ``gcn_norm`` defaults) as a dense fp32 matrix on ``device``."""
    A = torch.zeros(n, n, dtype=torch.float32)
    if edges:
        e = torch.tensor([p for p in edges if p[0] != p[1]], dtype=torch.long).view(-1, 2)
        A[e[:, 0], e[:, 1]] = 1.0
        A[e[:, 1], e[:, 0]] = 1.0
    A += torch.eye(n)
    dinv = A.sum(dim=1).pow(-0.5)
    return (dinv[:, None] * A * dinv[None, :]).to(device)


def _pad4(a_norm):
    """Zero columns up to a multiple of 4: the GEMM wants 16-byte rows (the matching rows of X W^T are zero-padded too)."""
    n = a_norm.shape[1]
    return a_norm if n % 4 == 0 else torch.nn.functional.pad(a_norm, (0, 4 - n % 4)).contiguous()


@torch.no_grad()
def gnn_pool(gnn, feats, a_norm):
    """mean over nodes of ``GNN(feats)`` -> [d]: per layer ``A_norm (X W^T) + b`` = two GEMMs on the HIP library."""
    x = feats
    a_pad = _pad4(a_norm)                                             # [n, n4]
    for i, conv in enumerate(gnn.convs):
        xw = _linear(x, conv.lin)                                     # X W^T              [n, out]
        xw = torch.nn.functional.pad(xw, (0, 0, 0, a_pad.shape[1] - xw.shape[0])).contiguous()
        x = ops.conv1d(a_pad, xw, conv.bias.contiguous())             # A_norm (X W^T) + b
        if i != gnn.n_layers - 1:
            x = torch.relu(x)
    return x.mean(dim=0)


@torch.no_grad()
def fusion_rows(args, model, tokenizer, dataset, idxs_sim, top_k):
    """The rows spliced in after position 2 (``H_sim_``): [1, d] for graphpooling, [m, d] for mlp."""
    wte = model.transformer.wte.weight
    idxs = [int(v) for v in idxs_sim][:top_k]
    if args.fusion == "graphpooling":
        order, edges = star_union_graph(dataset.retrieval_sources, idxs)
        feats = wte[torch.tensor(order, dtype=torch.long, device=wte.device)]
        return gnn_pool(model.gnn_fusion, feats, gcn_norm_dense(len(order), edges, wte.device)).view(1, -1)
    if args.fusion == "mlp":
        max_len_sim = 512
        cat = []
        for n in idxs:
            cat += list(dataset.retrieval_sources[n])
        cat = cat[:max_len_sim] + [tokenizer.pad_token_id] * max(0, max_len_sim - len(cat))
        h = wte[torch.tensor(cat, dtype=torch.long, device=wte.device)]         # [512, d]
        h = h.contiguous().view(-1, max_len_sim)                                # flat reinterpretation, utils/model.py:156
        mods = list(model.mlp_fusion.layers)
        for j, mod in enumerate(mods):
            if isinstance(mod, nn.Linear):
                h = _linear(h, mod)
            else:
                h = torch.relu(h)
        return h.contiguous().view(args.m, -1)                                  # :158 (batch of one)
    raise ValueError(f"unknown fusion {args.fusion!r} (mlp | graphpooling)")


@torch.no_grad()
def fusion_host_prep(args, model, dataset, index_lists, top_k):
    """Host half of ``fusion_rows_batch`` for one-layer graph pooling: the union-of-stars graphs of all queries and their
    pooling weights ``c`` (no GPU work -- the evaluation runs it for batch i+1 on a helper thread while batch i
    decodes).  Returns (node ids, C [n_queries, n_nodes] float32) or None for the other fusion configurations."""
    gnn = getattr(model, "gnn_fusion", None)
    if not (args.fusion == "graphpooling" and gnn is not None and gnn.n_layers == 1):
        return None
    nodes_all, spans, weights = [], [], []
    for ix in index_lists:
        order, edges = star_union_graph(dataset.retrieval_sources, [int(v) for v in ix][:top_k])
        n = len(order)
        e = np.asarray([p for p in edges if p[0] != p[1]], dtype=np.int64).reshape(-1, 2)
        deg = np.ones(n, dtype=np.float64)
        np.add.at(deg, e[:, 0], 1.0)
        np.add.at(deg, e[:, 1], 1.0)
        dinv = deg ** -0.5
        acc = dinv.copy()                                              # self loop
        np.add.at(acc, e[:, 0], dinv[e[:, 1]])
        np.add.at(acc, e[:, 1], dinv[e[:, 0]])
        weights.append(dinv * acc / n)
        spans.append((len(nodes_all), n))
        nodes_all += order
    ntot = (len(nodes_all) + 3) // 4 * 4                               # 16-byte GEMM rows
    C = np.zeros((len(index_lists), ntot), dtype=np.float32)
    for q, ((o, n), c) in enumerate(zip(spans, weights)):
        C[q, o:o + n] = c
    nodes_all += [0] * (ntot - len(nodes_all))                         # padding nodes carry weight 0
    return np.asarray(nodes_all, dtype=np.int64), C


@torch.no_grad()
def fusion_rows_batch(args, model, tokenizer, dataset, index_lists, top_k, prep=None):
    """``fusion_rows`` of many queries -> [n, r, d].  One-layer graph pooling (the shipped configuration) collapses
    algebraically: mean_i (A_norm (X W^T) + b)_i = sum_j c_j (X W^T)_j + b with c_j = (1/n) sum_i A_norm[i,j]
    = d_j^-1/2 (d_j^-1/2 + sum_{i in N(j)} d_i^-1/2) / n, so the whole batch is ONE gather, ONE projection GEMM over the
    concatenated nodes and ONE [n_queries, n_nodes] x [n_nodes, d] GEMM with the block-diagonal pooling weights -- no
    per-query adjacency matrices (``fusion_host_prep`` builds c; pass its result as ``prep`` when it was computed
    ahead).  Other configurations fall back to the per-query path."""
    if prep is None:
        prep = fusion_host_prep(args, model, dataset, index_lists, top_k)
    if prep is None:
        return torch.stack([fusion_rows(args, model, tokenizer, dataset, ix, top_k) for ix in index_lists])
    nodes, C = prep
    wte = model.transformer.wte.weight
    gnn = model.gnn_fusion
    X = wte[torch.from_numpy(nodes).to(wte.device)]
    conv = gnn.convs[0]
    Y = _linear(X, conv.lin)                                           # [ntot, d]
    out = ops.conv1d(torch.from_numpy(C).to(wte.device), Y.contiguous(), conv.bias.contiguous())
    return out.unsqueeze(1)


@torch.no_grad()
def fused_next_token(model, indexed_tokens, sim_rows):
    """argmax of the last position's logits of ``model(inputs_embeds=cat(H[:, :2], H_sim, H[:, 2:]))``
    (``utils/model.py:160-164,214-223``; ``Evaluation_generator.py:160``)."""
    wte = model.transformer.wte.weight
    H = wte[torch.tensor(indexed_tokens, dtype=torch.long, device=wte.device)]
    H_aug = torch.cat([H[:2], sim_rows, H[2:]], dim=0).unsqueeze(0).contiguous()
    hidden = model.transformer.encode(None, H_aug, want_hidden=True)["hidden"]
    logits = ops.lm_logits(hidden[:, -1, :].contiguous(), model.lm_head.weight)     # == wte when tied
    return int(torch.argmax(logits[0]).item())


@torch.no_grad()
def greedy_decode_rag(args, model, tokenizer, dataset, indexed_tokens, index, mode, max_len, n_spl):
    """Decode loop of ``get_eval_metrics_generator`` (:141-167) for one query."""
    sim_rows = fusion_rows(args, model, tokenizer, dataset, index, args.topK)    # constant over the steps
    toks = list(indexed_tokens)
    eos = tokenizer.encode("<|endoftext|>")
    gen_len = 0
    while True:
        nxt = fused_next_token(model, toks, sim_rows)
        toks.append(nxt)
        gen_len += 1
        if mode == "val":
            if gen_len > 10:
                break
        elif len(toks) >= max_len - n_spl:
            break
        if nxt in eos:
            break
    return toks


@torch.no_grad()
def greedy_decode_rag_batch(args, model, tokenizer, dataset, token_lists, index_lists, mode, max_len, n_spl, prep=None):
    """``greedy_decode_rag`` for MANY queries at once, with a key/value cache.

    The reference decodes one query at a time and re-runs the full forward for every generated token
    (``Evaluation_generator.py:141-167``).  The queries are independent and the model is causal, so here
      * the fused prompts of all queries are PREFILLED in a few length-grouped forwards that also fill the K/V cache
        (padding sits behind a sequence's real positions and can never reach them, so grouping changes nothing);
      * every further token is ONE ``r4d_gpt2_decode_step_f32`` over all queries: one new row per query against its
        cached keys/values (``modeling_gpt2.py:177-197`` ``layer_past`` semantics), lm_head on those rows only.
    Same tokens as the one-at-a-time loop up to fp32 summation order.  Finished queries keep their slot (their rows
    are ignored); argmax and the stop rules run on the device (``GreedyDecoder``), one captured graph per token."""
    if len(token_lists) == 0:
        return []
    return _rag_batch_finish(_rag_batch_start(args, model, tokenizer, dataset, token_lists, index_lists, mode, max_len, n_spl, prep))


@torch.no_grad()
def _rag_batch_start(args, model, tokenizer, dataset, token_lists, index_lists, mode, max_len, n_spl, prep=None, slot=0):
    """First half of ``greedy_decode_rag_batch``, everything up to the filled key/value cache: fusion rows, augmented prompt
    embeddings, length-grouped prefill.  All of it is queued on the CURRENT stream without a host wait; ``slot`` picks one
    of the decoders kept per batch shape, so that the next batch can be prefilled while this one still decodes."""
    tr = model.transformer
    wte = tr.wte.weight
    dev = wte.device
    n = len(token_lists)
    sims = fusion_rows_batch(args, model, tokenizer, dataset, index_lists, args.topK, prep)                 # [n, r, d]
    r = sims.shape[1]
    eos = tokenizer.encode("<|endoftext|>")
    toks = [list(t) for t in token_lists]
    lens0 = [len(t) + r for t in toks]                                 # augmented prompt lengths
    tmax = max(lens0)
    budget = 11 if mode == "val" else max(1, max_len - n_spl - min(len(t) for t in toks))
    cap = min(tmax + budget + 1, tr.wpe.num_embeddings)
    ids_h = np.zeros((n, tmax), dtype=np.int64)                        # augmented layout: [t0 t1 | r fused slots | t2 ...]
    for i, t in enumerate(toks):                                       # built on the host: ONE upload
        ids_h[i, :2] = t[:2]
        ids_h[i, 2 + r:len(t) + r] = t[2:]
    H_aug = wte[torch.from_numpy(ids_h).to(dev)]
    H_aug[:, 2:2 + r] = sims
    dec = tr.greedy_decoder(n, cap, slot)
    last = tr.prefill_last(dec.cache, lens0, inputs_embeds=H_aug)      # length-grouped forwards, cache rows [0, lens)
    lens = torch.tensor(lens0, dtype=torch.int32).to(dev, non_blocking=True)
    # stop rules of Evaluation_generator.py:168-175 in augmented positions (r fused rows sit inside every prompt)
    limit = cap if mode == "val" else min(cap, max_len - n_spl + r)
    return dict(dec=dec, last=last, lens=lens, toks=toks, max_gen=11 if mode == "val" else cap, limit=limit, eos=eos)


@torch.no_grad()
def _rag_batch_finish(st):
    """Second half: the device greedy loop on the CURRENT stream (waits for the generated ids), prompts + generated ids."""
    gen = st["dec"].run(st["last"], st["lens"], st["max_gen"], st["limit"], st["eos"])
    for t, g in zip(st["toks"], gen):
        t.extend(g)
    return st["toks"]


def decode_rag_batches(args, model, tokenizer, dataset, batches, mode, max_len, n_spl):
    """``greedy_decode_rag_batch`` over a list of (token_lists, index_lists) batches, yielding each batch's outputs.  The host
    half of the next batch's fusion (graph construction, pure Python / numpy) runs on a helper thread while the current batch
    decodes.  (A second form that also ran batch b + 1's fusion rows and prefill on a second HIP stream while batch b decoded was
    built in round 2, measured SLOWER -- UCI_13 shape, batch 32: 35.5 k against 38.2 k tokens/s: the prefill's GEMM workgroups hold
    every CU for tens of microseconds at a time and each of the decode step's 33 dependent launches queues behind them -- and
    removed in round 3.)"""
    from concurrent.futures import ThreadPoolExecutor
    if not batches:
        return

    def start(b, prep):
        toks, idxs = batches[b]
        return _rag_batch_start(args, model, tokenizer, dataset, toks, idxs, mode, max_len, n_spl, prep, slot=b % 2)

    with ThreadPoolExecutor(max_workers=1) as pool:
        nxt = start(0, pool.submit(fusion_host_prep, args, model, dataset, batches[0][1], args.topK).result())
        fut = pool.submit(fusion_host_prep, args, model, dataset, batches[1][1], args.topK) if len(batches) > 1 else None
        for b in range(len(batches)):
            st = nxt
            if b + 1 < len(batches):                                   # queue the next batch's fill BEFORE waiting on this decode
                prep = fut.result()
                if b + 2 < len(batches):
                    fut = pool.submit(fusion_host_prep, args, model, dataset, batches[b + 2][1], args.topK)
                nxt = start(b + 1, prep)                               # its decoder slot was released by finish(b - 1)
            yield _rag_batch_finish(st)


# ------------------------------------------------------------------------------------------------ dataset / eval
class TextIndexScoreDataset:
    """``dataloader/generator.py:12-80``: texts, retrieved index / score rows and the retrieval sources (train pool)."""

    def __init__(self, tokenizer, args, text_file_path, index_file_path, score_file_path, block_size=512):
        for p in (text_file_path, index_file_path, score_file_path):
            assert os.path.isfile(p), p
        train_data = read_nonblank_lines(args.train_data_file)
        text_lines = read_nonblank_lines(text_file_path)
        self.egolist, self.egoId = [], {}
        for i, line in enumerate(text_lines):
            ego = int(line.split('<|history|>')[1].split(' ')[1])
            self.egoId[ego] = i
            self.egolist.append(ego)
        from .retriever import read_matrix_rows              # binary side-car when this build wrote the files
        self.index = read_matrix_rows(index_file_path, int)
        self.score = read_matrix_rows(score_file_path, float)
        self.text = tokenizer(text_lines, add_special_tokens=True, max_length=block_size)["input_ids"]
        self.retrieval_sources = tokenizer(train_data, add_special_tokens=True, max_length=block_size)["input_ids"]

    def __len__(self):
        return len(self.text)

    def __getitem__(self, i):
        return (torch.tensor(self.text[i], dtype=torch.long), torch.tensor(self.index[i], dtype=torch.long),
                torch.tensor(self.score[i], dtype=torch.float), torch.tensor(self.egolist[i], dtype=torch.long))

    def __gethistory__(self, i):
        return torch.tensor(self.text[i], dtype=torch.long)

    def get_item_by_egoId(self, egoId):
        """``dataloader/generator.py:73-80``: the text of the LAST line whose ego id is ``egoId`` (None when absent)."""
        if egoId not in self.egoId or self.egoId[egoId] >= len(self.text):
            return None
        return self.__gethistory__(self.egoId[egoId])


def load_and_cache_examples(args, tokenizer, evaluate=False, test=False):
    """``dataloader/generator.py:83-101``."""
    if evaluate:
        paths = (args.eval_data_file, args.val_index_file, args.val_score_file)
    elif test:
        paths = (args.test_data_file, args.test_index_file, args.test_score_file)
    else:
        paths = (args.train_data_file, args.train_index_file, args.train_score_file)
    return TextIndexScoreDataset(tokenizer, args, *paths, block_size=args.block_size)


def get_eval_metrics_generator(args, epoch, model, tokenizer, step, mode="val", is_rag=False, is_best=False):
    """Drop-in for ``utils/Evaluation_generator.get_eval_metrics_generator`` (:49-265): R@5 / NDCG@5 / Jaccard of the
    greedy link predictions, ``eval_results.json`` and (``is_best``) the results CSV under ``rag_results/...``."""
    spl_tokens = tokenizer.additional_special_tokens + [tokenizer.bos_token, tokenizer.eos_token, tokenizer.pad_token]
    if mode == 'val':
        files = (args.eval_data_file, args.eval_data_gt_file, args.val_score_file, args.val_index_file)
    else:
        files = (args.test_data_file, args.test_data_gt_file, args.test_score_file, args.test_index_file)
    data, data_gt, data_score, data_index = (read_nonblank_lines(f) for f in files)
    assert len(data) == len(data_gt)
    with open(os.path.join('./vocabs', args.dataset, str(args.timestamp), 'vocab.json')) as f:
        vocab = json.load(f)
    indicator, root_path = ('do_train', 'rag_results/train_mode') if args.do_train else ('do_val', 'rag_results/val_mode')
    out_dir = os.path.join(root_path, args.dataset, str(args.timestamp), args.run_name,
                           "results_seed" if args.run_seed else "results", mode + '_score')
    os.makedirs(out_dir, exist_ok=True)

    Eval = Evaluation()
    model.eval()
    MAX_LEN = model.config.n_ctx
    topk = [5]
    metric_terms = ['R', 'NDCG', 'jaccard']
    top_k_scores = {metric: len(topk) * [0] for metric in metric_terms}
    generated_dict = {}
    train_dataset = load_and_cache_examples(args, tokenizer, evaluate=False)
    num_user_test = 0
    jobs = []                                           # (i, input_text, user_id, target_list, ids, index, num_user_test)
    for i, (input_text, text_gt, index, _score) in enumerate(zip(data, data_gt, data_index, data_score)):
        index = list(map(int, index.split()))
        generated_dict[i] = {}
        user_id = input_text.split()[2]
        target_list = [t for t in text_gt.split()[1:-2] if t != user_id and t in vocab]
        if len(target_list) == 0:
            print('text_gt: ', text_gt)
            continue
        indexed_tokens = tokenizer.encode(input_text)
        num_user_test += 1
        if len(indexed_tokens) > MAX_LEN:
            print('len_input: ', len(indexed_tokens))
            indexed_tokens = indexed_tokens[-1000:]
        jobs.append((i, input_text, user_id, target_list, indexed_tokens, index, num_user_test))
    bs = max(1, int(getattr(args, "per_gpu_eval_batch_size", 32) or 32))
    chunks = [jobs[b0:b0 + bs] for b0 in range(0, len(jobs), bs)]      # the queries are independent: a batch per step
    # data-parallel over the test queries (one process per GPU, main_generator.py under torch.distributed.run): a rank
    # decodes every world-th batch, the generated ids are all-gathered as objects, and EVERY rank then scores all queries
    # in file order -- the sums are those of the single-process run, bit for bit
    import torch.distributed as dist
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
    mine = chunks[rank::world]
    if is_rag:
        results = decode_rag_batches(args, model, tokenizer, train_dataset,
                                     [([j[4] for j in c], [j[5] for j in c]) for c in mine], mode, MAX_LEN,
                                     len(spl_tokens))
    else:
        from .evaluation import greedy_decode
        dev0 = next(model.parameters()).device
        results = ([greedy_decode(model, tokenizer, j[4], mode, MAX_LEN, len(spl_tokens), dev0) for j in c] for c in mine)
    generated = {}
    for chunk, outs in zip(mine, results):
        for job, out_ids in zip(chunk, outs):
            generated[job[0]] = [int(t) for t in out_ids]
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, generated)
        generated = {k: v for part in parts for k, v in part.items()}
    for (i, input_text, user_id, target_list, indexed_tokens, _ix, nut) in jobs:
        out_ids = generated[i]
        predicted_list = tokenizer.decode(out_ids).split()[len(indexed_tokens):]
        predicted = [t for t in predicted_list if t != user_id and t not in spl_tokens]
        for topi, k in enumerate(topk):
            try:
                top_k_scores['NDCG'][topi] += Eval.ndcg_k(predicted, target_list, k)
            except ZeroDivisionError:
                pass
            top_k_scores['jaccard'][topi] += Eval.jaccard(predicted, target_list)
            top_k_scores['R'][topi] += Eval.recall_k(predicted, target_list, k)
        generated_dict[i].update({'user_id': user_id, 'input': input_text, 'target_list': target_list,
                                  'len input_text': len(input_text.split()), 'predicted_list_ori': predicted_list,
                                  'predicted': predicted, 'NDCG@k': str(Eval.ndcg_k(predicted, target_list, 1)),
                                  'num_user_test': str(nut)})
    for metric in metric_terms:
        for topi, _k in enumerate(topk):
            top_k_scores[metric][topi] = round(top_k_scores[metric][topi] / max(num_user_test, 1), 4)
    if rank != 0:                                       # rank 0 writes the files
        return top_k_scores
    result_save_file = os.path.join(out_dir, mode + '_results_epoch.csv')
    if is_best:
        with open(result_save_file, 'w') as f:
            f.write('epoch,' + ''.join(p + ',' for p in args.para_names))
            f.write(''.join(f'R@{k},' for k in topk) + ''.join(f'NDCG@{k},' for k in topk) +
                    ''.join(f'jaccard@{k},' for k in topk) + '\n')
            f.write(str(epoch) + ',' + ''.join(str(v) + ',' for v in args.para_values))
            for metric in metric_terms:
                f.write(''.join(str(top_k_scores[metric][j]) + ',' for j in range(len(topk))))
            f.write('\n')
    with open(out_dir + '/eval_results.json', 'wt') as f:
        json.dump(generated_dict, f, indent=4)
    if is_best:
        import pandas as pd
        save_folder = os.path.join(indicator, mode + ('_metrics_seed_gen' if args.run_seed else '_metrics_ft_gen'))
        os.makedirs(save_folder, exist_ok=True)
        result_save_test = os.path.join(save_folder, args.dataset + '_SimpleDyG.csv')
        test_results = pd.read_csv(result_save_file)
        if os.path.exists(result_save_test):
            test_results.to_csv(result_save_test, mode='a', header=False, index=False)
        else:
            test_results.to_csv(result_save_test, index=False)
    return top_k_scores
