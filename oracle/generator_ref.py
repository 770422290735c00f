"""torch-CPU restatement of the RAG generator's inference path (oracle; test infra only) -- SURVEY.md section 8f-1.

Follows ``utils/model.py:105-224`` (``fusion_mlp`` / ``fusion_graphpooling``), ``models/modeling_rag.py:44-99``
(``GNN`` / ``MLP_custom``) and the decode loop of ``utils/Evaluation_generator.py:141-167``.

PARITY NOTE -- partly unpinned.  ``GCNConv`` and ``from_networkx`` live in torch_geometric (README.md:11 pins
``torch_geometric>=1.7.2``), which is neither in the reference tree nor installed here.  ``gcn_norm_dense`` /
``gcn_conv`` restate the published algorithm of ``GCNConv(in, out)`` with its defaults (``improved=False``,
``add_self_loops=True``, ``normalize=True``, bias): remove existing self loops, add exactly one of weight 1 per node
(``add_remaining_self_loops``), ``deg_i = sum_j w_ji``, ``norm_ij = deg_i^-1/2 w_ij deg_j^-1/2``, ``out = A_norm (x W^T) + b``.
Everything around it IS pinned: the union-of-stars graph and its node order against networkx itself
(``tests/test_oracle_golden.py``), the embedding splice, the flat ``view`` reshapes of ``fusion_mlp`` and the decode
loop against the reference's own GPT-2 forward through ``gpt2_ref`` (golden-pinned).
"""
import torch

from . import gpt2_ref


def make_mlp_state(seed, input_dim, output_dim, n_layers, std=0.05):
    """Deterministic ``MLP_custom`` state dict (keys ``layers.<i>.{weight,bias}``, ``modeling_rag.py:74-99``): the golden
    generator loads exactly these tensors into the reference module, so they are not committed."""
    g = torch.Generator().manual_seed(seed)
    hidden = int(input_dim / 2)
    dims = [(input_dim, output_dim)] if n_layers == 1 else \
        [(input_dim, hidden)] + [(hidden, hidden)] * (n_layers - 2) + [(hidden, output_dim)]
    sd = {}
    for i, (a, b) in enumerate(dims):
        j = i if n_layers == 1 else 2 * i                 # Sequential indices: Linear, ReLU, Linear, ...
        sd[f"layers.{j}.weight"] = torch.randn(b, a, generator=g) * std
        sd[f"layers.{j}.bias"] = torch.randn(b, generator=g) * std
    return sd


def mlp_layers_of(sd):
    """[(weight, bias), ...] in layer order from a ``make_mlp_state`` / reference ``MLP_custom`` state dict."""
    idx = sorted({int(k.split(".")[1]) for k in sd if k.startswith("layers.")})
    return [(sd[f"layers.{j}.weight"], sd[f"layers.{j}.bias"]) for j in idx]


def star_union_graph(retrieval_sources, idxs):
    """The fused graph of ``fusion_graphpooling`` (``utils/model.py:181-189``): for every retrieved sequence, edges
    from its ego node (token at position 2) to EVERY token id of the sequence (special tokens and the ego itself
    included -> one self loop), accumulated in an undirected ``nx.Graph``.  Returns (node ids in networkx insertion
    order, set of undirected edges as sorted index pairs, self loops included)."""
    order, pos, edges = [], {}, set()

    def node(v):
        if v not in pos:
            pos[v] = len(order)
            order.append(v)
        return pos[v]

    for n in idxs:
        seq = [int(e) for e in retrieval_sources[int(n)]]
        ego = int(seq[2])
        for e in seq:                                   # add_edges_from([(ego, elem) ...]): u first, then v
            a, b = node(ego), node(e)
            edges.add((min(a, b), max(a, b)))
    return order, edges


def gcn_norm_dense(n, edges):
    """Dense ``D^-1/2 (A + I) D^-1/2`` of GCNConv's ``gcn_norm`` for an undirected, unweighted graph."""
    A = torch.zeros(n, n, dtype=torch.float32)
    for a, b in edges:
        if a != b:
            A[a, b] = 1.0
            A[b, a] = 1.0
    A = A + torch.eye(n)
    dinv = A.sum(dim=1).pow(-0.5)
    return dinv[:, None] * A * dinv[None, :]


def gcn_conv(x, a_norm, weight, bias):
    """``GCNConv.forward``: ``weight`` is ``lin.weight`` [out, in]."""
    return a_norm @ (x @ weight.t()) + bias


def gnn_forward(x, a_norm, convs):
    """``GNN.forward`` (``modeling_rag.py:65-71``) for ``convs = [(weight, bias), ...]``.  With more than one layer
    the reference applies ``F.dropout`` with its default ``training=True`` even at inference (stochastic); the
    restatement uses the expectation (no dropout) and the product documents the deviation.  Shipped scripts use one layer."""
    for i, (w, b) in enumerate(convs):
        x = gcn_conv(x, a_norm, w, b)
        if i != len(convs) - 1:
            x = torch.relu(x)
    return x


def mlp_custom_forward(x, layers):
    """``MLP_custom.forward`` (``modeling_rag.py:74-99``): Linear (+ReLU) stack, ``layers = [(weight [out,in], bias), ...]``."""
    for i, (w, b) in enumerate(layers):
        x = x @ w.t() + b
        if i != len(layers) - 1:
            x = torch.relu(x)
    return x


@torch.no_grad()
def fusion_graphpooling_embeds(sd, retrieval_sources, tokens, idxs, top_k, convs):
    """``H_aug`` of ``fusion_graphpooling`` (``utils/model.py:167-219``) for ONE query: [1, T+1, d]."""
    wte = sd["transformer.wte.weight"]
    order, edges = star_union_graph(retrieval_sources, list(idxs)[:top_k])
    feats = wte[torch.tensor(order, dtype=torch.long)]
    emb = gnn_forward(feats, gcn_norm_dense(len(order), edges), convs)
    h_sim = emb.mean(dim=0).view(1, 1, -1)
    H = wte[torch.tensor([list(tokens)], dtype=torch.long)]
    return torch.cat([H[:, :2], h_sim, H[:, 2:]], dim=1)


@torch.no_grad()
def fusion_mlp_embeds(sd, retrieval_sources, tokens, idxs, top_k, m, layers, pad_id, max_len_sim=512):
    """``H_aug`` of ``fusion_mlp`` (``utils/model.py:105-164``) for ONE query: [1, T+m, d].  The two ``view`` calls
    reinterpret the flat buffer (no transpose), exactly as upstream."""
    wte = sd["transformer.wte.weight"]
    d = wte.shape[1]
    cat = []
    for n in list(idxs)[:top_k]:
        cat += [int(e) for e in retrieval_sources[int(n)]]
    cat = cat[:max_len_sim] + [pad_id] * max(0, max_len_sim - len(cat))
    h_sim = wte[torch.tensor([cat], dtype=torch.long)]                  # [1, 512, d]
    h_sim = h_sim.contiguous().view(-1, h_sim.size(1))                  # [d, 512]  (flat reinterpretation)
    h_sim = mlp_custom_forward(h_sim, layers).contiguous().view(-1, m, d)
    H = wte[torch.tensor([list(tokens)], dtype=torch.long)]
    return torch.cat([H[:, :2], h_sim, H[:, 2:]], dim=1)


@torch.no_grad()
def greedy_decode_rag(sd, n_head, embeds_fn, indexed_tokens, eos_id, mode="val", max_len=1024, n_spl=0, eps=1e-5):
    """Batch-1 greedy decode of ``utils/Evaluation_generator.py:141-167``: every step re-fuses (``embeds_fn(tokens)``
    -> H_aug) and re-runs the full forward on ``inputs_embeds``; argmax of the last position."""
    toks = list(indexed_tokens)
    gen_len = 0
    while True:
        out = gpt2_ref.gpt2_forward(sd, None, n_head, eps, inputs_embeds=embeds_fn(toks))
        nxt = int(torch.argmax(out["logits"][0, -1, :]).item())
        toks.append(nxt)
        gen_len += 1
        if mode == "val":
            if gen_len > 10:
                break
        elif len(toks) >= max_len - n_spl:
            break
        if nxt == eos_id:
            break
    return toks
