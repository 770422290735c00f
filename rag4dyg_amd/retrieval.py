"""Encode -> score -> rank on the GPU: the device half of the retriever's ``test()``.

Reference: ``train/train_retriever.py:414-443`` (pool / query encode loops, normalise, dot, (x+1)/2) and
``:357-358,461-467`` (ranking).  All arithmetic runs in the HIP library; this module only batches.
"""
import torch

from . import ops


def right_pad_batches(examples, batch_size, pad_id, device):
    """Sequential batches right-padded to the batch max with [PAD], drop_last=False
    (``dataloader/retriever.py:153-166``).  Returns a list of int64 [b,T] device tensors."""
    out = []
    for s in range(0, len(examples), batch_size):
        chunk = examples[s:s + batch_size]
        T = max(len(e) for e in chunk)
        b = torch.full((len(chunk), T), int(pad_id), dtype=torch.int64)
        for i, e in enumerate(chunk):
            b[i, :len(e)] = torch.as_tensor(e, dtype=torch.int64)
        out.append(b.to(device, non_blocking=True))
    return out


MAX_FUSED_ROWS = 65536          # token rows per fused launch sequence (bounds the workspace: ~20 KB per row at d=512)


def _encode_batches_once(model, batches, max_rows):
    out, group, rows = [], [], 0
    for b in batches:
        r = b.shape[0] * b.shape[1]
        if group and rows + r > max_rows:
            out.append(model.encode_groups_meanpool(group))
            group, rows = [], 0
        group.append(b)
        rows += r
    if group:
        out.append(model.encode_groups_meanpool(group))
    return torch.cat(out, dim=0)


@torch.no_grad()
def encode_batches(model, batches, max_rows=MAX_FUSED_ROWS, check=True):
    """HOT LOOP 1/2: mean-pooled embeddings of every batch, concatenated (``train_retriever.py:414-422,430-432``).

    Consecutive batches are handed to the library in fused groups of up to ``max_rows`` token rows; every batch
    keeps its own padding, so the values equal the one-batch-per-call reference loop.

    Range guard (``include/r4d.h``, ABI v6; ``check=False`` leaves it to the caller's ``ops.take_range_flag``): after the last
    group ONE read of the device's range word.  If a non-finite hidden state came up in gemm mode "f16x2" (an activation beyond
    fp16's exponent range) the WHOLE call is re-run once under "bf16x3" -- fp32's range, the same accuracy -- with a warning;
    if it comes up again, or in any other mode, ``R4DError``: NaN embeddings are never returned."""
    flag_word = ops.range_flag(batches[0].device if batches else None)
    if check:
        flag_word.zero_()                                             # (somebody else's unread bits are not this call's)
    emb = _encode_batches_once(model, batches, max_rows)
    if not check:
        return emb
    flag = ops.take_range_flag()
    if flag & ops.RANGE_NONFINITE_HIDDEN and ops.gemm_mode() == "f16x2":
        import warnings
        warnings.warn("rag4dyg_amd: an activation left the fp16 range of the f16x2 arithmetic (non-finite hidden state); "
                      "re-encoding this call with the bf16x3 GEMMs", RuntimeWarning, stacklevel=2)
        ops.set_gemm_mode("bf16x3")
        try:
            emb = _encode_batches_once(model, batches, max_rows)
            flag = ops.take_range_flag()
        finally:
            ops.set_gemm_mode("f16x2")
    if flag & ops.RANGE_NONFINITE_HIDDEN:
        raise ops._lib.R4DError("encode_batches: a non-finite hidden state reached ln_f (weights or inputs overflow fp32 itself, "
                                "or hold NaN): no embeddings returned")
    return emb


class PoolIndex:
    """Row-normalised pool shard resident in HBM (the reference re-normalises and re-uploads the pool for
    every query batch, ``train_retriever.py:435-436``; the values are identical)."""

    def __init__(self, pool_emb, index_offset=0, check=True):
        self.pool_hat = ops.normalize_rows(pool_emb.contiguous())
        self.index_offset = int(index_offset)
        if check:                                                    # a NaN / zero pool row never becomes part of an index
            ops.check_range("PoolIndex")

    def __len__(self):
        return self.pool_hat.shape[0]

    def search(self, q_emb, k, want_scores=False, check=True):
        """-> (vals [Q,k], global idx int64 [Q,k], scores [Q,N] | None); canonical (score desc, index asc) order.
        ``check`` (default): one read of the range-guard word before the result is handed out -- a query row that is NaN / inf /
        zero (or an unread non-finite encode before it) raises ``R4DError`` instead of returning the top-k of NaN scores
        (the selection orders NaN below everything: an ordinary-looking list).  ``check=False`` keeps the call asynchronous
        (pipelined callers read ``ops.take_range_flag`` at their own synchronisation point)."""
        q_hat = ops.normalize_rows(q_emb.contiguous())
        out = ops.score_topk(q_hat, self.pool_hat, k, self.index_offset, want_scores)
        if check:
            ops.check_range("PoolIndex.search")
        return out

    def search_hat(self, q_hat, k, offset=None):
        v, i, _ = ops.score_topk(q_hat, self.pool_hat, k, self.index_offset if offset is None else offset)
        return v, i
