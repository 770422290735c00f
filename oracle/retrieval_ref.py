"""CPU restatement of the retriever's encode -> score -> rank path (oracle; test infra only).

Follows ``train/train_retriever.py:376-483`` (``test``) and the batching of
``dataloader/retriever.py:128-168``.
"""
import numpy as np
import torch

from . import gpt2_ref


def right_pad_batches(examples, batch_size, pad_id):
    """Sequential batches, right-padded with ``pad_id`` to the batch max, ``drop_last=False``.

    ``dataloader/retriever.py:153-166`` (eval branch: ``SequentialSampler`` +
    ``pad_sequence(batch_first=True, padding_value=pad_token_id)``).
    """
    out = []
    for s in range(0, len(examples), batch_size):
        chunk = examples[s:s + batch_size]
        T = max(len(e) for e in chunk)
        b = torch.full((len(chunk), T), pad_id, dtype=torch.long)
        for i, e in enumerate(chunk):
            b[i, :len(e)] = torch.as_tensor(e, dtype=torch.long)
        out.append(b)
    return out


@torch.no_grad()
def encode_batches(sd, n_head, batches, eps=1e-5):
    """HOT LOOP 1/2 -- ``train/train_retriever.py:414-422,430-432``.

    ``_, h = model(input_ids); h.mean(dim=1)`` -- the mean runs over ALL padded
    positions (no attention_mask is passed, :419), so an embedding depends on
    the batch it was padded with.
    """
    embs = []
    for ids in batches:
        h = gpt2_ref.gpt2_forward(sd, ids, n_head, eps, want_logits=False)["hidden"]
        embs.append(torch.mean(h, dim=1))
    return torch.cat(embs, dim=0)


@torch.no_grad()
def score_batch(q_emb, pool_emb):
    """``train/train_retriever.py:433-438``: row-normalise both sides (no eps),
    ``S = q_hat @ p_hat.T``, ``S = (S + 1) / 2``."""
    qn = q_emb / q_emb.norm(dim=1, keepdim=True)
    pn = pool_emb / pool_emb.norm(dim=1, keepdim=True)
    return (torch.matmul(qn, pn.t()) + 1) / 2


def rank_full(scores, stable=True):
    """``save_index_score`` -- ``train/train_retriever.py:357-358``: ``np.argsort(-S, axis=1)``.

    The reference's default argsort is unstable; the canonical tie-break used
    by the build (and by every parity check) is ascending pool index, i.e.
    ``kind='stable'`` (SURVEY.md section 7 hard-part 1).
    """
    s = np.asarray(scores)
    return np.argsort(-s, axis=1, kind="stable" if stable else None)


def topk_stable(scores, k):
    """Top-k (value desc, index asc) of each row -> (values, indices)."""
    idx = rank_full(scores)[:, :k]
    return np.take_along_axis(np.asarray(scores), idx, axis=1), idx


def hit_rate_at_k(predictions, targets, k=1):
    """``train/train_retriever.py:31-38``."""
    gt = set(int(t) for t in targets)
    return 1 if any(int(i) in gt for i in predictions[:k]) else 0


def hit_metrics(score_batches, gt_batches, stable=True):
    """hit@1 / hit@3 exactly as accumulated at ``train/train_retriever.py:458-479``:
    per-batch mean, then mean over batches, rounded to 4 places; the ground
    truth top-3 comes from the float32 Jaccard rows (:409-410, :461-462)."""
    hit1 = hit3 = 0.0
    steps = 0
    kind = "stable" if stable else None
    for S, G in zip(score_batches, gt_batches):
        S = np.asarray(S)
        G = np.asarray(G, dtype=np.float32)
        hb1 = hb3 = 0
        for i in range(S.shape[0]):
            gt = np.argsort(-G[i], kind=kind)[:3]
            pred = np.argsort(-S[i], kind=kind)
            hb1 += hit_rate_at_k(pred, gt, 1)
            hb3 += hit_rate_at_k(pred, gt, 3)
        hit1 += hb1 / S.shape[0]
        hit3 += hb3 / S.shape[0]
        steps += 1
    return round(hit1 / steps, 4), round(hit3 / steps, 4)


def bce_with_logits_mean(scores, gt):
    """``nn.BCEWithLogitsLoss()(dot_products, score)`` -- ``train/train_retriever.py:439-441``."""
    return torch.nn.functional.binary_cross_entropy_with_logits(
        torch.as_tensor(scores, dtype=torch.float32), torch.as_tensor(gt, dtype=torch.float32)).item()


def format_index_rows(indices):
    """Text of ``*_index.gen`` rows -- ``train/train_retriever.py:361-362``."""
    return [" ".join(str(int(x)) for x in row) for row in indices]


def format_score_rows(scores):
    """Text of ``*_score.gen`` rows (``%.4f``) -- ``train/train_retriever.py:363``."""
    return [" ".join(f"{x:.4f}" for x in row) for row in np.asarray(scores)]


def topk_matches_modulo_ties(ref_scores, got_idx, k, tol):
    """Tie-aware top-k check used by parity tests.

    ``got_idx[r,:k]`` is accepted for row r iff every returned index has a
    reference score >= (k-th best reference score - tol) and the returned
    indices are distinct -- i.e. the selection can differ from the reference
    only inside a band of width ``tol`` around the rank-k boundary.
    """
    ref_scores = np.asarray(ref_scores, dtype=np.float64)
    for r in range(ref_scores.shape[0]):
        kth = np.sort(ref_scores[r])[::-1][k - 1]
        sel = np.asarray(got_idx[r, :k])
        if len(set(sel.tolist())) != k:
            return False
        if (ref_scores[r, sel] < kth - tol).any():
            return False
    return True
