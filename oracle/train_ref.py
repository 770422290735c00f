"""torch-CPU restatement of ONE retriever training step (oracle; test infra only) -- SURVEY.md section 8f-4.

Follows ``train/train_retriever.py:40-98`` (``CLtime_loss``, ``mask_correlated_samples``, ``info_nce``), ``:177-196`` (the five
forwards and the loss of ``train_epoch``) and ``models/modeling_rag.py:774-840`` (``_aug``).  Pinned by
``tests/golden/g8_training_step.npz`` (the reference's own functions and autograd, run by ``oracle/gen_golden.py g8``).
Dropout is the identity in the fixtures (generated with all dropout probabilities 0: the reference's p = 0.1 masks come from
the device RNG and cannot be reproduced on another device); ``PhiloxDropout`` restates the PRODUCT's mask generator so that a
training-mode step can be checked given the same masks.

PARITY NOTE -- the optimizer update is unpinned.  ``adamw_step`` / ``clip_coefficient`` restate ``transformers.AdamW.step``
(the class ``utils/model.py:80-93`` instantiates; third-party, README.md:9 ``transformers>=4.24.0``) and
``torch.nn.utils.clip_grad_norm_``.  ``transformers.AdamW`` does not exist in the installed transformers 5.15 (removed upstream)
and is not in the reference tree, so the restatement follows its published algorithm -- bias-corrected step size
``lr * sqrt(1 - b2^t) / (1 - b1^t)``, denominator ``sqrt(v) + eps`` (eps NOT bias-corrected), decoupled decay
``p -= lr * wd * p`` applied AFTER the update -- and is cross-checked against ``torch.optim.AdamW``, an independent
implementation of the same moments, in the one regime where the two coincide exactly (no decay, torch's eps rescaled by
``1 / sqrt(1 - b2^t)``; ``tests/test_oracle_golden.py``).  Loss values, embeddings and gradients ARE pinned (G8).
"""
import math
import random

import numpy as np
import torch
import torch.nn.functional as F

from . import gpt2_ref


def cltime_loss(temperature, decay_rate, anchors, positives, hard_negatives, anchors_time, positives_time, negatives_time):
    """``CLtime_loss`` -- ``train/train_retriever.py:40-72``: cosine similarities of [anchors; positives; hard negatives],
    each block scaled by exp(-lambda |t_anchor - t_other|) (in-batch anchors: diagonal zeroed), one cross entropy with the
    positive of the same row as the label."""
    B = anchors.size(0)
    allv = torch.cat([anchors, positives, hard_negatives], dim=0)
    sim = F.cosine_similarity(allv.unsqueeze(1), allv.unsqueeze(0), dim=2)
    d_pos = torch.exp(-decay_rate * torch.abs(anchors_time.unsqueeze(1) - positives_time).squeeze())
    d_neg = torch.exp(-decay_rate * torch.abs(anchors_time.unsqueeze(1) - anchors_time).squeeze())
    d_neg.fill_diagonal_(0)
    d_hard = torch.exp(-decay_rate * torch.abs(anchors_time.unsqueeze(1) - negatives_time).squeeze())
    logits = torch.cat([sim[:B, B:2 * B] * d_pos, sim[:B, :B] * d_neg, sim[:B, 2 * B:] * d_hard], dim=1) / temperature
    return F.cross_entropy(logits, torch.arange(B))


def mask_correlated_samples(batch_size):
    """``train/train_retriever.py:74-82``."""
    N = 2 * batch_size
    mask = torch.ones((N, N), dtype=bool)
    mask = mask.fill_diagonal_(0)
    for i in range(batch_size):
        mask[i, batch_size + i] = 0
        mask[batch_size + i, i] = 0
    return mask


def info_nce(z_i, z_j, temp, batch_size):
    """``info_nce`` -- ``train/train_retriever.py:84-98`` (plain dot products, NOT normalised)."""
    N = 2 * batch_size
    z = torch.cat((z_i, z_j), dim=0)
    sim = torch.mm(z, z.T) / temp
    pos = torch.cat((torch.diag(sim, batch_size), torch.diag(sim, -batch_size)), dim=0).reshape(N, 1)
    neg = sim[mask_correlated_samples(batch_size)].reshape(N, -1)
    return F.cross_entropy(torch.cat((pos, neg), dim=1), torch.zeros(N, dtype=torch.long))


def aug(batch_seqs, eta, gamma, mask_token):
    """``_aug`` -- ``models/modeling_rag.py:774-840``: first view = crop, second view = mask, driven by python's ``random``
    (seed it to reproduce).  Quirks kept: the length of a row is its count of NON-ZERO ids (pads count, node id 0 does
    not); the crop copies a window of ``floor(length * eta)`` ids to the END of an all-zero row; ``seq[:]`` of an ndarray
    is a view, so the mask view is written into the row itself (after the crop view was taken)."""
    seqs = batch_seqs.tolist()
    lengths = batch_seqs.count_nonzero(dim=1).tolist()
    out1, out2 = [], []
    for seq, length in zip(seqs, lengths):
        seq = np.asarray(list(seq), dtype=np.int64)

        def crop():
            num_left = math.floor(length * eta)
            crop_begin = random.randint(4, length - num_left)
            c = np.zeros_like(seq)
            if crop_begin != 0:
                c[-num_left:] = seq[-(crop_begin + num_left):-crop_begin]
            else:
                c[-num_left:] = seq[-(crop_begin + num_left):]
            return c.tolist(), num_left

        def mask():
            num_mask = math.floor(length * gamma)
            idx = random.sample(range(length), k=num_mask)
            m = seq[:]
            m[[-i - 1 for i in idx]] = mask_token
            return m.tolist(), length

        if length > 1:
            a1, l1 = crop()
            out1.append(a1 if l1 > 0 else seq.tolist())
            a2, l2 = mask()
            out2.append(a2 if l2 > 0 else seq.tolist())
        else:
            out1.append(seq.tolist()); out2.append(seq.tolist())
    return torch.tensor(out1, dtype=torch.long), torch.tensor(out2, dtype=torch.long)


def training_step(sd, n_head, anchor, pos, neg, all_times, idx, eta, gamma, alpha, temperature, decay_rate, mask_token,
                  seed, with_grad=False, drop=None):
    """The loss of one ``train_epoch`` iteration (``train/train_retriever.py:177-196``): three forwards + time-decayed
    contrastive loss, two augmented forwards + alpha * InfoNCE.  ``with_grad``: ``sd`` tensors must require grad; the
    returned loss is then differentiable (gradient fixtures)."""
    fwd = gpt2_ref.gpt2_forward.__wrapped__ if with_grad else gpt2_ref.gpt2_forward
    def emb(ids):                                         # ``drop``: a PhiloxDropout (training mode), told each batch's shape
        if drop is not None:
            drop.next_group(ids.shape[0], ids.shape[1])
        return fwd(sd, ids, n_head, want_logits=False, drop=drop)["hidden"].mean(dim=1)
    h_a, h_p, h_n = emb(anchor), emb(pos), emb(neg)
    t = torch.as_tensor(all_times)
    cl = cltime_loss(temperature, decay_rate, h_a, h_p, h_n, t[idx[:, 0:1]], t[idx[:, 1:2]], t[idx[:, 2:3]])
    random.seed(seed)
    a1, a2 = aug(anchor, eta, gamma, mask_token)
    h1, h2 = emb(a1), emb(a2)
    au = alpha * info_nce(h1, h2, temperature, a1.size(0))
    return dict(cl=cl, aug=au, loss=cl + au, emb=torch.stack([h_a, h_p, h_n, h1, h2]), aug1=a1, aug2=a2)


def clip_coefficient(grads, max_norm):
    """``torch.nn.utils.clip_grad_norm_`` (``train/train_retriever.py:210``): min(1, max_norm / (total L2 norm + 1e-6))."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).item()
    return min(1.0, max_norm / (total + 1e-6)), total


def adamw_step(p, g, m, v, t, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """One update of ``transformers.AdamW`` (the optimizer ``utils/model.py:80-93`` builds; third-party code that is not in the
    reference tree -- transformers >= 4.24, README.md:9 -- restated from its published ``step``): bias correction on, the
    denominator is sqrt(v) + eps WITHOUT bias correction, decoupled weight decay applied after the update with the plain
    learning rate.  Returns the new (p, m, v); ``t`` is the 1-based update count."""
    b1, b2 = betas
    m = b1 * m + (1.0 - b1) * g
    v = b2 * v + (1.0 - b2) * g * g
    step_size = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    p = p - step_size * m / (v.sqrt() + eps)
    if weight_decay > 0.0:
        p = p - lr * weight_decay * p
    return p, m, v


# ------------------------------------------------------------------------------------------------ dropout masks
def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox-4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) on numpy uint64 arrays
    holding 32-bit words: ten rounds of (hi, lo) = M * c, key bumped by the Weyl constants after every round.  The device
    generator (``rag4dyg_amd/csrc/train_ops.hip``) is this function."""
    import numpy as np
    M0, M1, W0, W1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0) & MASK, np.uint64(k1) & MASK
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> np.uint64(32)) ^ c3 ^ k1) & MASK, p0 & MASK
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c0, c1, c2, c3


def philox_keep(n, p, seed, step, site, base=0):
    """Keep-mask (bool [n]) of dropout probability ``p`` for elements base .. base + n - 1 of ``site`` at ``step``: element e takes
    word e % 4 of the Philox block with counter (e // 4 low word, site + (e // 4 high word << 16), step low, step high) and key
    (seed low, seed high); kept iff word >= floor(p * 2^32)."""
    import numpy as np
    assert n % 4 == 0 and base % 4 == 0
    blk = np.arange(base // 4, (base + n) // 4, dtype=np.uint64)
    site_w = np.uint64(site) + ((blk >> np.uint64(32)) << np.uint64(16))
    w = philox4x32_10(blk & np.uint64(0xFFFFFFFF), site_w, np.uint64(step & 0xFFFFFFFF), np.uint64(step >> 32),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    words = np.stack(w, axis=1).reshape(-1)
    return words >= np.uint64(int(p * 4294967296.0))


class PhiloxDropout:
    """The dropout modules of the training forward with the product's mask generator and element numbering (``include/r4d.h``
    r4d_train_dropout): [rows, d] activations are numbered over the concatenated rows of a step's batches in call order,
    attention probabilities over the [B*H, T, ceil128(T)] blocks laid end to end.  ``next_group(B, T)`` before each forward."""
    SITE_EMBD = 65535

    def __init__(self, embd_p, attn_p, resid_p, seed, step):
        self.p = {"embd": embd_p, "attn": attn_p, "resid_attn": resid_p, "resid_mlp": resid_p}
        self.seed, self.step = seed, step
        self.row_next = self.p_next = 0
        self.row0 = self.p0 = 0
        self.masks = {}

    def next_group(self, B, T):
        self.row0, self.p0 = self.row_next, self.p_next
        self.row_next += B * T
        self._BT = (B, T)

    def __call__(self, kind, layer, x):
        p = self.p[kind]
        if p <= 0:
            return x
        if kind == "attn":
            B, H, T, _ = x.shape
            ld = (T + 127) // 128 * 128
            if layer == 0:
                self.p_next = self.p0 + B * H * T * ld
            keep = philox_keep(B * H * T * ld, p, self.seed, self.step, 4 * layer, self.p0).reshape(B, H, T, ld)[..., :T]
        else:
            site = self.SITE_EMBD if kind == "embd" else 4 * layer + (1 if kind == "resid_attn" else 2)
            keep = philox_keep(x.numel(), p, self.seed, self.step, site, self.row0 * x.shape[-1]).reshape(tuple(x.shape))
        m = torch.from_numpy(keep).to(x.dtype) / (1.0 - p)
        return x * m
