"""torch-CPU fp32 restatement of the SimpleDyG GPT-2 forward (oracle; test infra only).

Functional style over a reference-layout ``state_dict`` (keys as written by
``models/modeling_utils.py:277-297`` ``save_pretrained``):

    transformer.wte.weight [V,d]      transformer.wpe.weight [n_positions,d]
    transformer.h.<i>.ln_1.{weight,bias} [d]
    transformer.h.<i>.attn.c_attn.{weight [d,3d], bias [3d]}
    transformer.h.<i>.attn.c_proj.{weight [d,d],  bias [d]}
    transformer.h.<i>.ln_2.{weight,bias} [d]
    transformer.h.<i>.mlp.c_fc.{weight [d,4d],  bias [4d]}
    transformer.h.<i>.mlp.c_proj.{weight [4d,d], bias [d]}
    transformer.ln_f.{weight,bias} [d]       lm_head.weight (tied alias of wte)

Dropout is the identity (``model.eval()``, ``train/train_retriever.py:400``).
"""
import math

import torch


def conv1d(x, weight, bias):
    """``Conv1D.forward`` -- ``models/modeling_utils.py:1267-1271``.

    ``addmm(bias, x[BT,nx], W[nx,nf])``: weight is stored [in,out].
    """
    size_out = x.shape[:-1] + (weight.shape[1],)
    y = torch.addmm(bias, x.reshape(-1, x.shape[-1]), weight)
    return y.view(size_out)


def gelu_new(x):
    """``gelu_new`` -- ``models/modeling_gpt2.py:25,206`` (tanh form from
    ``transformers.activations``): 0.5x(1+tanh(sqrt(2/pi)(x+0.044715x^3)))."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def layer_norm(x, weight, bias, eps=1e-5):
    """``nn.LayerNorm(nx, eps=config.layer_norm_epsilon)`` --
    ``models/modeling_gpt2.py:219,221,339``; eps 1e-5 ``configuration_gpt2.py:131``."""
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), weight, bias, eps)


def attn_core(q, k, v, drop=None):
    """``Attention._attn`` -- ``models/modeling_gpt2.py:140-160`` with ``scale=True`` (:338).

    q [B,H,T,hd], k [B,H,hd,T], v [B,H,T,hd].  The mask is the reference's
    multiplicative/additive form ``w*b - 1e4*(1-b)`` (:146) -- masked logits are
    exactly -10000, NOT -inf -- followed by a softmax over all T keys (:152).
    """
    w = torch.matmul(q, k)
    w = w / math.sqrt(v.size(-1))
    nd, ns = w.size(-2), w.size(-1)
    b = torch.tril(torch.ones(ns, ns, dtype=w.dtype))[ns - nd:ns, :ns].view(1, 1, nd, ns)
    w = w * b - 1e4 * (1 - b)
    w = torch.softmax(w, dim=-1)
    if drop is not None:
        w = drop(w)                                       # attn_dropout, :153 (training mode only)
    return torch.matmul(w, v)


def attention(x, sd, prefix, n_head, drop=None):
    """``Attention.forward`` -- ``models/modeling_gpt2.py:177-197`` (no past, no masks)."""
    B, T, d = x.shape
    hd = d // n_head
    qkv = conv1d(x, sd[prefix + "c_attn.weight"], sd[prefix + "c_attn.bias"])
    q, k, v = qkv.split(d, dim=2)
    q = q.view(B, T, n_head, hd).permute(0, 2, 1, 3)      # split_heads :166-172
    k = k.view(B, T, n_head, hd).permute(0, 2, 3, 1)      # k=True -> [B,H,hd,T]
    v = v.view(B, T, n_head, hd).permute(0, 2, 1, 3)
    a = attn_core(q, k, v, drop)
    a = a.permute(0, 2, 1, 3).contiguous().view(B, T, d)  # merge_heads :161-164
    return conv1d(a, sd[prefix + "c_proj.weight"], sd[prefix + "c_proj.bias"])


def mlp(x, sd, prefix):
    """``MLP.forward`` -- ``models/modeling_gpt2.py:209-212``."""
    h = gelu_new(conv1d(x, sd[prefix + "c_fc.weight"], sd[prefix + "c_fc.bias"]))
    return conv1d(h, sd[prefix + "c_proj.weight"], sd[prefix + "c_proj.bias"])


def block(x, sd, i, n_head, eps, drop=None):
    """``Block.forward`` -- ``models/modeling_gpt2.py:224-235`` (pre-LN residual).  ``drop(kind, layer, tensor)`` (training
    mode only) stands for the three nn.Dropout modules of a block: "attn" (:153), "resid_attn" (:194), "resid_mlp" (:212)."""
    p = f"transformer.h.{i}."
    a = attention(layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], eps), sd, p + "attn.", n_head,
                  None if drop is None else (lambda w: drop("attn", i, w)))
    if drop is not None:
        a = drop("resid_attn", i, a)
    x = x + a
    m = mlp(layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], eps), sd, p + "mlp.")
    if drop is not None:
        m = drop("resid_mlp", i, m)
    return x + m


def n_layers_of(sd):
    n = 0
    while f"transformer.h.{n}.ln_1.weight" in sd:
        n += 1
    return n


@torch.no_grad()
def gpt2_forward(sd, input_ids, n_head, eps=1e-5, inputs_embeds=None, want_logits=True, want_layers=False, drop=None):
    """``GPT2Model.forward`` + ``GPT2LMHeadModel.forward`` --
    ``models/modeling_gpt2.py:400-509,583-603`` (== ``models/modeling_rag.py:455-564,664-687``).

    position ids are always 0..T-1 (:420-423); no token types; logits = ``lm_head(hidden)``
    (``modeling_gpt2.py:585``): ``lm_head.weight`` is the tied ``wte`` (``modeling_utils.py:165-181``) unless the
    state dict carries its own (the reference unties it: ``utils/tokenizer.py:56-66``, ``utils/model.py:71-78``).  Returns a dict with
    ``hidden`` = ln_f output [B,T,d], optional ``logits`` [B,T,V] and the
    per-layer residual stream ``layers`` (embedding output first).
    """
    wte = sd["transformer.wte.weight"]
    wpe = sd["transformer.wpe.weight"]
    if inputs_embeds is None:
        inputs_embeds = wte[input_ids]
    T = inputs_embeds.shape[1]
    x = inputs_embeds + wpe[torch.arange(T)].unsqueeze(0)
    if drop is not None:
        x = drop("embd", None, x)                         # self.drop, :337,427 (training mode only)
    layers = [x]
    for i in range(n_layers_of(sd)):
        x = block(x, sd, i, n_head, eps, drop)
        layers.append(x)
    h = layer_norm(x, sd["transformer.ln_f.weight"], sd["transformer.ln_f.bias"], eps)
    out = {"hidden": h}
    if want_logits:
        out["logits"] = torch.matmul(h, sd.get("lm_head.weight", wte).t())
    if want_layers:
        out["layers"] = layers
    return out


def lm_loss(logits, labels):
    """Shifted cross entropy -- ``models/modeling_gpt2.py:604-615``."""
    shift_logits = logits[..., :-1, :].contiguous()
    shift_labels = labels[..., 1:].contiguous()
    return torch.nn.functional.cross_entropy(shift_logits.view(-1, shift_logits.size(-1)), shift_labels.view(-1))


def make_state_dict(n_layer, n_embd, vocab, n_positions=1024, seed=1234, std=0.02, random_affine=False):
    """Deterministic reference-layout weights for tests and benches.

    The reference initialises Linear/Embedding/Conv1D ~ N(0, 0.02), zero biases,
    LayerNorm weight 1 / bias 0 (``models/modeling_gpt2.py:251-262``).  The RNG
    draw ORDER of the reference constructor is not reproduced; instead the
    golden generator loads exactly these tensors into the reference model
    (``load_state_dict``), so both sides see identical weights.  With
    ``random_affine`` biases / LN affine are randomised too so that parity tests
    exercise them (a trained checkpoint has non-trivial values there).
    """
    g = torch.Generator().manual_seed(seed)

    def nrm(*shape, s=std):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * s

    d = n_embd
    sd = {"transformer.wte.weight": nrm(vocab, d), "transformer.wpe.weight": nrm(n_positions, d)}
    for i in range(n_layer):
        p = f"transformer.h.{i}."
        for ln in ("ln_1", "ln_2"):
            sd[p + ln + ".weight"] = 1.0 + nrm(d, s=0.1) if random_affine else torch.ones(d)
            sd[p + ln + ".bias"] = nrm(d, s=0.1) if random_affine else torch.zeros(d)
        for name, (nx, nf) in (("attn.c_attn", (d, 3 * d)), ("attn.c_proj", (d, d)),
                               ("mlp.c_fc", (d, 4 * d)), ("mlp.c_proj", (4 * d, d))):
            sd[p + name + ".weight"] = nrm(nx, nf)
            sd[p + name + ".bias"] = nrm(nf, s=0.05) if random_affine else torch.zeros(nf)
    sd["transformer.ln_f.weight"] = 1.0 + nrm(d, s=0.1) if random_affine else torch.ones(d)
    sd["transformer.ln_f.bias"] = nrm(d, s=0.1) if random_affine else torch.zeros(d)
    sd["lm_head.weight"] = sd["transformer.wte.weight"]
    return sd


@torch.no_grad()
def greedy_decode(sd, n_head, indexed_tokens, eos_id, mode="val", max_len=1024, n_spl=0, eps=1e-5):
    """Batch-1 greedy decode of ``utils/Evaluation_SimpleDyG.py:120-145``: full forward on the growing
    sequence each step, argmax of the last position, stop rules of val (10 tokens) / test (length cap) / EOS."""
    toks = list(indexed_tokens)
    gen_len = 0
    while True:
        out = gpt2_forward(sd, torch.tensor([toks]), n_head, eps)
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


def stress_transform(sd):
    """TRAINED-GPT-2 STATISTICS on top of a checkpoint (fixture G11): a deterministic edit of a
    reference-layout state dict that gives it the features large trained GPT-2s show and an N(0, 0.02) initialisation does
    not -- a handful of LayerNorm gains 6-10x the rest, two `wte` outlier channels and a `wpe` offset in EVERY channel (residual
    rows whose mean is many times their spread: the LayerNorm variance is a small difference of large numbers), one output
    column of a `c_proj` of every block 20x (a massive-activation channel written into the residual stream) and a `c_attn`
    query/key bias that sharpens the softmax.  Returns a new dict (tensors cloned); the tied `lm_head.weight` follows `wte`.
    Nothing here comes from the reference: it is an input generator, applied identically before the reference model
    (oracle/gen_golden.py g11) and the HIP path (tests) load the weights."""
    out = {k: v.clone() for k, v in sd.items() if k != "lm_head.weight"}
    d = out["transformer.wte.weight"].shape[1]
    n = n_layers_of(out)

    def ch(*frac):                                      # channel picks that scale with d (d 128 -> e.g. 3, 17, 64, 101)
        return sorted({int(f * d) % d for f in frac})
    for i in range(n):
        p = f"transformer.h.{i}."
        out[p + "ln_1.weight"][ch(0.025, 0.135, 0.5, 0.79)] *= 8.0
        out[p + "ln_2.weight"][ch(0.04, 0.31, 0.6)] *= 6.0
        out[p + "mlp.c_proj.weight"][:, ch(0.09 + 0.2 * i)] *= 20.0
        out[p + "attn.c_proj.weight"][:, ch(0.55 + 0.1 * i)] *= 20.0
        b = out[p + "attn.c_attn.bias"]
        b[ch(0.07)[0]] += 2.0                           # one query channel ...
        b[d + ch(0.07)[0]] += 2.0                       # ... and the matching key channel
    out["transformer.ln_f.weight"][ch(0.017, 0.26, 0.7)] *= 10.0
    out["transformer.wte.weight"][:, ch(0.055, 0.39)] += 0.3
    out["transformer.wpe.weight"] += 0.25
    if "lm_head.weight" in sd and sd["lm_head.weight"].data_ptr() != sd["transformer.wte.weight"].data_ptr() \
            and not torch.equal(sd["lm_head.weight"], sd["transformer.wte.weight"]):
        out["lm_head.weight"] = sd["lm_head.weight"].clone()            # an untied head stays as it is
    else:
        out["lm_head.weight"] = out["transformer.wte.weight"]
    return out


def sharpen_attention(sd, factor):
    """PEAKED-SOFTMAX input generator (fixture G13): every block's `c_attn` query and key columns (weight
    and bias) x ``factor``, i.e. every attention logit x ``factor``^2 -- an N(0, 0.02) initialisation has logits of a few tenths
    at head_dim 128 / 256, a trained model several tens to hundreds.  Values are untouched.  Like ``stress_transform`` nothing
    here comes from the reference: it is applied identically before the reference model (oracle/gen_golden.py g13) and the HIP
    path (tests) load the weights.  Returns a new dict; a tied `lm_head.weight` stays the `wte` alias."""
    out = {k: v.clone() for k, v in sd.items() if k != "lm_head.weight"}
    d = out["transformer.wte.weight"].shape[1]
    for i in range(n_layers_of(out)):
        p = f"transformer.h.{i}.attn.c_attn."
        out[p + "weight"][:, :2 * d] *= float(factor)
        out[p + "bias"][:2 * d] *= float(factor)
    if "lm_head.weight" in sd and not torch.equal(sd["lm_head.weight"], sd["transformer.wte.weight"]):
        out["lm_head.weight"] = sd["lm_head.weight"].clone()
    else:
        out["lm_head.weight"] = out["transformer.wte.weight"]
    return out


def weight_bit_checksums(sd):
    """uint64 per tensor (sorted by name, the tied ``lm_head.weight`` left out): the sum of its fp32 BIT PATTERNS mod 2^64 -- an
    exact signature, the same on every machine (a floating-point sum of 10^6 elements depends on the thread count)."""
    import numpy as np
    return np.array([int(sd[k].detach().contiguous().numpy().view(np.uint32).astype(np.uint64).sum(dtype=np.uint64))
                     for k in sorted(sd) if k != "lm_head.weight"], dtype=np.uint64)
