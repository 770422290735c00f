"""Pool sharding across the GPUs of one node and the one collective the path needs.

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI).  The retrieval pool is
partitioned into contiguous, BATCH-ALIGNED index ranges (multiples of 32 rows): a pool embedding depends
on the 32-sequence batch it was padded with (train_retriever.py:420 averages over padded positions), so
only whole reference batches may move between ranks.  Per query batch the data path is

    all-gather query embeddings [Q_b, d]            (64 KB per rank at d=512)
    local scan of the rank's shard -> top-k          (no communication)
    all-gather (value f32, global index i64) [Q,k]   (3.8 KB per rank at Q=32, k=10)
    merge G*k candidates per row, (value desc, index asc)

Both collectives are latency-bound (tens of microseconds over xGMI), nowhere near the 153 GB/s/link
limit; the merged result is identical to the single-GPU top-k by construction.
The reference has no analogue (single process; DataParallel replicates the model, train_retriever.py:391-392).
"""
import torch
import torch.distributed as dist


def shard_bounds(n_rows, world_size, align=32):
    """Contiguous [start,end) per rank, starts aligned to `align` rows, sizes as even as the alignment allows."""
    n_units = (n_rows + align - 1) // align
    base, rem = divmod(n_units, world_size)
    bounds, start = [], 0
    for r in range(world_size):
        units = base + (1 if r < rem else 0)
        end = min(n_rows, start + units * align)
        bounds.append((start, end))
        start = end
    return bounds


def all_gather_cat(t, group=None):
    """Concatenate equally-shaped tensors of every rank along dim 0 (one all-gather)."""
    world = dist.get_world_size(group)
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    return out


def balanced_runs(costs, parts):
    """Cut a sequence of item costs into ``parts`` contiguous runs of about equal total cost: run r ends at the first
    item whose cumulative cost reaches (r+1)/parts of the total.  Returns [start, end) per run (runs may be empty)."""
    total, cum, bounds, start, i = float(sum(costs)), 0.0, [], 0, 0
    for r in range(parts):
        target = total * (r + 1) / parts
        while i < len(costs) and (cum < target or r == parts - 1):
            cum += costs[i]
            i += 1
        bounds.append((start, i))
        start = i
    return bounds


def all_gather_rows(t, counts, group=None):
    """Concatenate row blocks of DIFFERENT heights (``counts[r]`` rows on rank r) along dim 0: one all-gather of blocks
    padded to the tallest.  With the gloo backend (1-GPU rehearsal) the block is staged through host memory."""
    world = dist.get_world_size(group)
    tall = max(counts)
    buf = torch.zeros((tall,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    buf[:t.shape[0]] = t
    if dist.get_backend(group) == "gloo" and buf.is_cuda:
        out = all_gather_cat(buf.cpu(), group).to(t.device)
    else:
        out = all_gather_cat(buf, group)
    return torch.cat([out[r * tall:r * tall + counts[r]] for r in range(world)])


def encode_pool_sharded(encode, batches, group=None):
    """Pool embeddings with the ENCODE sharded over the ranks: rank r encodes a contiguous run of whole reference batches
    (a pool embedding depends on the batch it was padded with, so batches never split), then one all-gather hands every
    rank the full [N, d] matrix in pool order.  ``encode(list of batches) -> [rows, d]``."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    bounds = balanced_runs([int(b.shape[0]) * int(b.shape[1]) for b in batches], world)     # equal padded positions
    counts = [sum(int(b.shape[0]) for b in batches[s:e]) for s, e in bounds]
    s, e = bounds[rank]
    mine = encode(batches[s:e]) if e > s else None
    if mine is None:
        ref = encode(batches[:1])                                        # a rank without work still needs d and the dtype
        mine = ref[:0]
    return all_gather_rows(mine, counts, group)


def pack_candidates(vals, idx):
    """(vals [Q,k] f32, idx [Q,k] i64) -> ONE uint8 tensor [1, Q, 12k] (the value bytes, then the index bytes, per query): the
    per-shard candidates travel in one all-gather instead of two (the collectives of a step are
    latency-bound, far from the per-link bandwidth).  Lossless: ``unpack_candidates`` returns the same bits."""
    Q = vals.shape[0]
    return torch.cat([vals.contiguous().view(torch.uint8).reshape(Q, -1), idx.contiguous().view(torch.uint8).reshape(Q, -1)],
                     dim=1).unsqueeze(0)


def unpack_candidates(packed, k):
    """[G, Q, 12k] uint8 (``pack_candidates`` of G ranks) -> (vals [G,Q,k] f32, idx [G,Q,k] i64)."""
    G, Q = packed.shape[0], packed.shape[1]
    vals = packed[..., :4 * k].contiguous().view(torch.float32).reshape(G, Q, k)
    idx = packed[..., 4 * k:].contiguous().view(torch.int64).reshape(G, Q, k)
    return vals, idx


def sharded_topk(q_hat_all, pool_hat_shard, shard_offset, k, local_topk, merge, group=None):
    """Global top-k of every query against the sharded pool.

    local_topk(q_hat, pool_hat, k, offset) -> (vals [Q,k] f32, idx [Q,k] i64)   (HIP: ops.score_topk)
    merge(vals [G,Q,k], idx [G,Q,k]) -> (vals [Q,k], idx [Q,k])               (HIP: ops.merge_topk)
    Ranks whose shard holds fewer than k rows contribute (-inf, INT64_MAX) padding.
    """
    Q = q_hat_all.shape[0]
    n_local = pool_hat_shard.shape[0]
    kk = min(k, n_local)
    vals = torch.full((Q, k), float("-inf"), dtype=torch.float32, device=q_hat_all.device)
    idx = torch.full((Q, k), torch.iinfo(torch.int64).max, dtype=torch.int64, device=q_hat_all.device)
    if kk > 0:
        v, i = local_topk(q_hat_all, pool_hat_shard, kk, shard_offset)
        vals[:, :kk] = v
        idx[:, :kk] = i
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return vals, idx
    gv, gi = unpack_candidates(all_gather_cat(pack_candidates(vals, idx), group), k)     # ONE collective: [G,Q,12k] bytes
    return merge(gv, gi)


class PipelinedShardedTopK:
    """``sharded_topk`` as a three-stage software pipeline over query batches (one batch per ``submit``).

    A batch needs two collectives (query embeddings, then the per-shard candidates: values and indices packed into one tensor).  Run synchronously, every rank waits for
    the slowest rank TWICE per batch -- and ranks are unequal from batch to batch, because each pads its own query batches
    to their own lengths.  Here both collectives are started asynchronously and consumed one ``submit`` later each:

        submit(i):  merge(i-2)  <- candidates gathered during step i-1
                    local scan + top-k of batch i-1 on the gathered queries, START the candidate all-gathers
                    START the all-gather of batch i's query embeddings

    so a rank only stalls when it is a whole step ahead of the slowest one.  ``submit`` returns the merged (vals, idx) of
    the batch submitted two calls earlier (``None`` while the pipeline fills); ``flush()`` drains it and returns the
    remaining results in order.  Same collectives in the same order on every rank; results identical to ``sharded_topk``.
    """

    def __init__(self, pool_hat_shard, shard_offset, k, local_topk, merge, group=None):
        self.pool, self.offset, self.k = pool_hat_shard, shard_offset, k
        self.local_topk, self.merge, self.group = local_topk, merge, group
        self.world = dist.get_world_size(group)
        self._q = None          # (work, gathered queries, keep-alive)
        self._cand = None       # (work, gathered packed candidates [G,Q,12k], keep)

    def _start_gather(self, t):
        t = t.contiguous()
        if t.is_cuda and dist.get_backend(self.group) == "gloo":       # 1-GPU rehearsal: staged through host memory
            th = t.cpu()
            oh = torch.empty((self.world * th.shape[0],) + tuple(th.shape[1:]), dtype=th.dtype)
            work = dist.all_gather_into_tensor(oh, th, group=self.group, async_op=True)

            class _Staged:                                             # wait(), then the result back on the device
                def wait(_self):
                    work.wait()
                    out.copy_(oh)
            out = torch.empty(oh.shape, dtype=t.dtype, device=t.device)
            return _Staged(), out, (t, th, oh)
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        work = dist.all_gather_into_tensor(out, t, group=self.group, async_op=True)
        return work, out, t

    def _step(self, q_hat):
        done = None
        if self._cand is not None:
            wc, gc, _ = self._cand
            wc.wait()
            done = self.merge(*unpack_candidates(gc, self.k))           # [G,Q,k] each
            self._cand = None
        if self._q is not None:
            w, q_all, _ = self._q
            w.wait()
            Q, n_local = q_all.shape[0], self.pool.shape[0]
            kk = min(self.k, n_local)
            vals = torch.full((Q, self.k), float("-inf"), dtype=torch.float32, device=q_all.device)
            idx = torch.full((Q, self.k), torch.iinfo(torch.int64).max, dtype=torch.int64, device=q_all.device)
            if kk > 0:
                v, i = self.local_topk(q_all, self.pool, kk, self.offset)
                vals[:, :kk] = v
                idx[:, :kk] = i
            self._cand = self._start_gather(pack_candidates(vals, idx))
            self._q = None
        if q_hat is not None:
            self._q = self._start_gather(q_hat)
        return done

    def submit(self, q_hat):
        return self._step(q_hat)

    def flush(self):
        out = []
        for _ in range(2):
            r = self._step(None)
            if r is not None:
                out.append(r)
        return out
