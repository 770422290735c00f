"""Jaccard pool annotation (``retrieval_data_annotation.py``) with the N x N work on the MI355X.

Host side: the reference's string parsing (J1/J2), the token -> dense-id map, CSR packing and the text file
writers in the reference's formats.  Device side: ``r4d_jaccard_f64`` (all four matrices), ``r4d_topk_f64``
(train top-10), ``r4d_argsort_desc_f64`` (full rankings, canonical stable order).
"""
import os

import numpy as np
import torch

from . import ops

ROW_CHUNK = 2048           # A-rows per device pass: bounds the f64 slab to ROW_CHUNK x N

# ---- phase timing (R4D_PHASE_TIMING=1): where the wall clock of an annotation run goes -- parse, CSR pack + upload, device
# work (Jaccard, rankings, selections), download, text formatting + file writes.  Off by default (the syncs it adds are the
# only cost); tools/annotation_e2e.py prints the table next to the CPU oracle's time.
import contextlib
import time

PHASES = {}
_TIMING = os.environ.get("R4D_PHASE_TIMING") == "1"


@contextlib.contextmanager
def phase(name):
    if not _TIMING:
        yield
        return
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    try:
        yield
    finally:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        PHASES[name] = PHASES.get(name, 0.0) + time.perf_counter() - t0



def get_input_seq(seq):
    """``retrieval_data_annotation.py:17-20`` -- keeps the ego id and the <|timeK|> tokens."""
    seq = seq.split('<|history|>')[1].split('<|endofhistory|>')[0].split(' ')
    return list(filter(lambda x: x != '', seq))


def get_output_seq(seq):
    """``retrieval_data_annotation.py:22-26`` -- drops anything containing 'time'."""
    seq = seq.split('<|pre|>')[1].split('<|endofpre|>')[0].split(' ')
    seq = list(filter(lambda x: x != '', seq))
    return list(filter(lambda x: 'time' not in x, seq))


def get_inout_list(data, gt):
    """``retrieval_data_annotation.py:28-34``."""
    return [get_input_seq(d) for d in data], [get_output_seq(g) for g in gt]


class SetTable:
    """Token-string lists -> device CSR (sorted unique dense ids).  One shared token map per annotation run."""

    def __init__(self):
        self.vocab = {}

    def pack(self, seqs, device):
        ptr = np.zeros(len(seqs) + 1, dtype=np.int32)
        rows = []
        for i, s in enumerate(seqs):
            row = sorted({self.vocab.setdefault(t, len(self.vocab)) for t in s})
            rows.append(row)
            ptr[i + 1] = ptr[i] + len(row)
        idx = np.fromiter((x for r in rows for x in r), dtype=np.int32, count=int(ptr[-1]))
        if idx.size == 0:
            idx = np.zeros(1, dtype=np.int32)
        return torch.from_numpy(ptr).to(device), torch.from_numpy(idx).to(device)


def _slice_csr(ptr, idx, r0, r1):
    p = ptr[r0:r1 + 1]
    base = int(p[0].item())
    end = int(p[-1].item())
    sub_idx = idx[base:max(end, base + 1)].contiguous()
    return (p - base).contiguous(), sub_idx


def occurrence_matrix(target_csr, source_csr, vocab, zero_diag=False):
    """Device f64 [len(target), len(source)] == ``occurrence_matrix`` (``retrieval_data_annotation.py:36-41``)
    (+ ``np.fill_diagonal(.., 0)`` :172-173 when ``zero_diag``)."""
    return ops.jaccard(target_csr[0], target_csr[1], source_csr[0], source_csr[1], vocab, zero_diag)


def rank_row_range(n_rows, rank, world):
    """Contiguous target rows [start, end) of ``rank``: rows of the target side are independent, so they
    are dealt to the ranks in order -- the source-side CSR is replicated and there is no collective on the data path."""
    base, rem = divmod(n_rows, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def iter_row_chunks(target_csr, source_csr, vocab, zero_diag=False, chunk=ROW_CHUNK, rows=None):
    """Yield (row0, f64 slab [rows, N]) so that an N x N matrix never has to exist at once.

    With ``zero_diag`` the slab is computed against the full source and the diagonal entries of the global
    matrix (row0+i, row0+i) are zeroed on device.  ``rows = (start, end)`` restricts the walk to that range of target rows
    (one rank's share).
    """
    n = target_csr[0].numel() - 1
    lo, hi = (0, n) if rows is None else rows
    for r0 in range(lo, hi, chunk):
        r1 = min(hi, r0 + chunk)
        with phase("device: jaccard"):
            p, ix = _slice_csr(target_csr[0], target_csr[1], r0, r1)
            m = ops.jaccard(p, ix, source_csr[0], source_csr[1], vocab, False)
            if zero_diag:
                ar = torch.arange(r1 - r0, device=m.device)
                m[ar, ar + r0] = 0.0
        yield r0, m


# ---- text: the reference's files are O(Q x N) decimal numbers.  ``str`` per number was 0.24 s of a 0.45 s hepth run; a Jaccard
# matrix holds a few hundred DISTINCT values (quotients of small integers) and the indices are 0 .. N-1, so every number's text is
# made once and the rows are assembled by table lookup (same bytes: the table entries ARE ``str(float)`` / ``str(int)``).
_INT_STR = np.empty(0, dtype=object)


def _int_strings(n):
    """object array: _int_strings(n)[i] == str(i) for i < n (grown on demand, shared by every file of a run)."""
    global _INT_STR
    if _INT_STR.size < n:
        _INT_STR = np.array([str(i) for i in range(max(n, 2 * _INT_STR.size))], dtype=object)
    return _INT_STR


def _float_lines(m):
    """Rows of a device f64 matrix as text, ``' '.join(str(x) for x in row)`` per row: distinct values found on the device
    (``torch.unique``), their shortest round-trip text made once, rows assembled by lookup."""
    u, inv = torch.unique(m, return_inverse=True)
    tab = np.array([str(x) for x in u.cpu().tolist()], dtype=object)
    inv = inv.to(torch.int32 if u.numel() < 2 ** 31 else torch.int64).cpu().numpy()
    return [' '.join(tab[r].tolist()) for r in inv]


def _int_lines(idx_np):
    tab = _int_strings(int(idx_np.max()) + 1 if idx_np.size else 1)
    return [' '.join(tab[r].tolist()) for r in idx_np]


def save_index_score(target_csr, source_csr, vocab, save_index_file, save_score_file, rows=None):
    """``retrieval_data_annotation.py:88-93``: full ranking (canonical stable order) + every score, as text."""
    with open(save_index_file, 'w') as f, open(save_score_file, 'w') as g:
        for _r0, m in iter_row_chunks(target_csr, source_csr, vocab, rows=rows):
            with phase("device: full-row argsort"):
                perm = ops.argsort_desc(m)
            with phase("download"):
                indices = perm.cpu().numpy()
            with phase("text: format + write"):
                f.write('\n'.join(_int_lines(indices)) + '\n')
                g.write('\n'.join(_float_lines(m)) + '\n')


def save_score_file_train(out_csr, vocab, save_file_index, save_file_score, topk=10, rows=None):
    """``retrieval_data_annotation.py:97-103``: top-10 GT demonstrations of train x train (diag zeroed)."""
    with open(save_file_index, 'w') as f_index, open(save_file_score, 'w') as f_score:
        for _r0, m in iter_row_chunks(out_csr, out_csr, vocab, zero_diag=True, rows=rows):
            with phase("device: top-10"):
                vals, idx = ops.topk_f64(m, min(topk, m.shape[1]))
            with phase("download"):
                idx_h = idx.cpu().numpy()
            with phase("text: format + write"):
                f_index.write('\n'.join(_int_lines(idx_h)) + '\n')
                f_score.write('\n'.join(_float_lines(vals)) + '\n')


def _mix64(x):
    """splitmix64 finaliser on int64 tensors (two's-complement wrap-around is the arithmetic wanted): a counter-based hash, so the
    pick of (anchor row i, its j-th positive) is a pure function of (seed, i, j) -- no generator state walks from row to row."""
    x = (x ^ (x >> 30 & 0x3FFFFFFFF)) * -4658895280553007687          # 0xBF58476D1CE4E5B9
    x = (x ^ (x >> 27 & 0x1FFFFFFFFF)) * -7723592293110705685         # 0x94D049BB133111EB
    return x ^ (x >> 31 & 0x1FFFFFFFF)


def train_annotation_chunks(out_csr, in_csr, vocab, threshold=0.8, neg_num=5, dialog=False, rows=None, choice_seed=0):
    """Per chunk of anchor rows, everything ``save_train_annotation`` writes, found ON THE DEVICE and downloaded as five small
    arrays: (anchor i, positive, picked negative, score of the positive, score of the negative) per (anchor, positive) pair in
    the reference's order (``retrieval_data_annotation.py:52-85``: anchors ascending, positives ascending; dialog keeps the first
    four positives, :73-74), plus -- for the tests -- the candidate lists.  Candidate order = canonical stable ranking of the
    input-set similarities (the reference's order is tie-dependent): walk it, first the non-positive entries
    with an output score > 0, then -- only if fewer than ``neg_num`` -- those with score 0.  The negative of a pair is a uniform
    pick among the candidates, as upstream's unseeded ``np.random.choice`` (:79), drawn by ``_mix64(seed, i, j)``."""
    gen_in = iter_row_chunks(in_csr, in_csr, vocab, zero_diag=True, rows=rows)
    for (r0, m_out), (_r0b, m_in) in zip(iter_row_chunks(out_csr, out_csr, vocab, zero_diag=True, rows=rows), gen_in):
        with phase("device: positives + argsort of the input-set rows"):
            posmask = m_out > threshold
            rows_d = torch.nonzero(posmask.any(dim=1)).flatten()
            if rows_d.numel() == 0:
                continue
            mo = m_out[rows_d]
            order_d = ops.argsort_desc(m_in[rows_d].contiguous()).long()
            ro = torch.gather(mo, 1, order_d)                                # output scores in ranking order
            notpos = ~(ro > threshold)
            cls = torch.where(notpos & (ro > 0), 0, torch.where(notpos & (ro == 0), 1, 2))
            npool = ro.shape[1]
            key = cls * npool + torch.arange(npool, device=ro.device)[None, :]
            kk = min(neg_num, npool)
            best = torch.topk(key, kk, dim=1, largest=False, sorted=True).values
            neg_idx_d = torch.gather(order_d, 1, best % npool)               # [rows, kk] candidates, in walk order
            neg_cnt_d = (best < 2 * npool).sum(dim=1)
            pr, pc = torch.nonzero(posmask[rows_d], as_tuple=True)           # (row, positive) pairs, row-major = the reference's order
            first = torch.searchsorted(pr, torch.arange(rows_d.numel(), device=pr.device))
            ordinal = torch.arange(pr.numel(), device=pr.device) - first[pr]  # j: which positive of its row
            if dialog:
                keep = ordinal < 4
                pr, pc, ordinal = pr[keep], pc[keep], ordinal[keep]
            anchor = rows_d[pr] + r0
            ncand = neg_cnt_d[pr]
            if bool((ncand == 0).any()):
                raise ValueError("'a' cannot be empty unless no samples are taken")     # np.random.choice([]) upstream
            h = _mix64(_mix64(anchor + int(choice_seed)) + ordinal)
            pick = torch.remainder(h & 0x7FFFFFFFFFFFFFFF, ncand)
            neg = neg_idx_d[pr, pick]
            s_pos, s_neg = mo[pr, pc], mo[pr, neg]
        with phase("download"):
            out = dict(anchor=anchor.cpu().numpy(), pos=pc.cpu().numpy(), neg=neg.cpu().numpy(), s_pos=s_pos, s_neg=s_neg,
                       cand_rows=(rows_d + r0).cpu().numpy(), cand_idx=neg_idx_d.cpu().numpy(), cand_cnt=neg_cnt_d.cpu().numpy())
        yield out


def save_train_annotation(out_csr, in_csr, vocab, save_file, save_file_score, threshold=0.8, neg_num=5, dataset="",
                          rows=None, choice_seed=None):
    """``retrieval_data_annotation.py:43-85``: ``i pos neg`` triples; the negative is a uniform random pick among the
    candidates as upstream (``np.random.choice``, :79 -- unseeded there, so column 3 is not reproducible by design).  Here the
    pick is a counter-based hash of (``choice_seed``, anchor row, which positive) -- ``choice_seed`` itself is drawn from numpy's
    global state, so ``np.random.seed`` still governs a run: a row's picks do not depend on which rows were processed before it,
    which is what lets the rows be dealt to several ranks and still give the one-rank file."""
    if choice_seed is None:
        choice_seed = int(np.random.randint(0, 2 ** 31 - 1))
    cnt, n = 0, out_csr[0].numel() - 1
    with open(save_file, 'w') as f, open(save_file_score, 'w') as g:
        for c in train_annotation_chunks(out_csr, in_csr, vocab, threshold, neg_num, 'dialog' in dataset, rows, choice_seed):
            with phase("text: format + write"):
                k = len(c["anchor"])
                if k == 0:
                    continue
                tab = _int_strings(int(max(c["anchor"].max(), c["pos"].max(), c["neg"].max())) + 1)
                a_s = tab[c["anchor"]].tolist()
                sc = _float_lines(torch.stack([c["s_pos"], c["s_neg"]], dim=1))
                f.write('\n'.join(map(' '.join, zip(a_s, tab[c["pos"]].tolist(), tab[c["neg"]].tolist()))) + '\n')
                g.write('\n'.join(a + ' ' + s_ for a, s_ in zip(a_s, sc)) + '\n')
                cnt += k
    return n, cnt


def read_lines(path):
    with open(path, 'r') as f:
        return [line for line in f.read().splitlines() if (len(line) > 0 and not line.isspace())]


def _join_parts(path, world):
    """Concatenate ``path.part<r>`` (r = 0..world-1, in rank order = row order) into ``path`` and remove the parts."""
    with open(path, 'wb') as out:
        for r in range(world):
            with open(f"{path}.part{r}", 'rb') as f:
                while True:
                    buf = f.read(1 << 24)
                    if not buf:
                        break
                    out.write(buf)
            os.remove(f"{path}.part{r}")


def main(argv):
    """``python retrieval_data_annotation.py <dataset> <timestamp> <threshold>`` (``:109-200``), cwd-relative paths.

    One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, e.g. by ``python -m torch.distributed.run``): the
    TARGET rows of each of the four matrices (``:166-169``) are dealt to the ranks in contiguous ranges, every rank holds
    the whole source-side CSR and writes its rows to ``<file>.part<rank>``; rank 0 concatenates the parts in rank order.
    No collective on the data path -- the process group (gloo) only carries one seed broadcast and the barrier before
    the concatenation; the files equal the one-process run byte for byte."""
    dataset, timestamp = argv[1], argv[2]
    threshold = float(argv[3])
    if not torch.cuda.is_available():
        raise SystemExit("retrieval_data_annotation: needs the MI355X (no CPU fallback in rag4dyg_amd)")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as tdist
    if world > 1 and not tdist.is_initialized():
        tdist.init_process_group(backend="gloo")
    save_path = os.path.join('./resources/', dataset, str(timestamp), 'train_retrieval')
    os.makedirs(save_path, exist_ok=True)
    save_path_gen = os.path.join('./resources/train_generator', dataset, str(timestamp), "train_gt_topk")
    os.makedirs(save_path_gen, exist_ok=True)
    base = os.path.join('resources', dataset, timestamp)
    _t_parse = time.perf_counter()
    train_data = read_lines(os.path.join(base, 'train.link_prediction'))
    test_data = read_lines(os.path.join(base, 'test.link_prediction'))
    test_gt = read_lines(os.path.join(base, 'test_gt.link_prediction'))
    val_data = read_lines(os.path.join(base, 'val.link_prediction'))
    val_gt = read_lines(os.path.join(base, 'val_gt.link_prediction'))
    train_in, train_out = get_inout_list(train_data, train_data)
    _, test_out = get_inout_list(test_data, test_gt)
    _, val_out = get_inout_list(val_data, val_gt)
    if _TIMING:
        PHASES["host: read + parse the five text files"] = time.perf_counter() - _t_parse
    table = SetTable()
    with phase("host: token map + CSR pack + upload"):
        tr_out = table.pack(train_out, device)
        tr_in = table.pack(train_in, device)
        te_out = table.pack(test_out, device)
        va_out = table.pack(val_out, device)
    vocab = len(table.vocab)
    seed = [int(np.random.randint(0, 2 ** 31 - 1))]
    if world > 1:
        tdist.broadcast_object_list(seed, src=0)                     # one seed for the random negative picks of every rank
    part = (lambda p: f"{p}.part{rank}") if world > 1 else (lambda p: p)
    rng_tr = rank_row_range(len(train_out), rank, world) if world > 1 else None
    files = {k: os.path.join(save_path, k) for k in ('train_index.retrieval', 'train_score.retrieval', 'test_index.retrieval',
                                                      'test_score.retrieval', 'val_index.retrieval', 'val_score.retrieval')}
    files.update({k: os.path.join(save_path_gen, k) for k in ('train_index.gen', 'train_score.gen')})
    n, cnt = save_train_annotation(tr_out, tr_in, vocab, part(files['train_index.retrieval']), part(files['train_score.retrieval']),
                                   threshold=threshold, neg_num=5, dataset=dataset, rows=rng_tr, choice_seed=seed[0])
    save_index_score(te_out, tr_out, vocab, part(files['test_index.retrieval']), part(files['test_score.retrieval']),
                     rows=rank_row_range(len(test_out), rank, world) if world > 1 else None)
    save_index_score(va_out, tr_out, vocab, part(files['val_index.retrieval']), part(files['val_score.retrieval']),
                     rows=rank_row_range(len(val_out), rank, world) if world > 1 else None)
    save_score_file_train(tr_out, vocab, part(files['train_index.gen']), part(files['train_score.gen']), topk=10, rows=rng_tr)
    if world > 1:
        counts = [None] * world
        tdist.all_gather_object(counts, cnt)                         # also the barrier: every part file is closed
        cnt = sum(counts)
        if rank == 0:
            for p in files.values():
                _join_parts(p, world)
        tdist.barrier()
    if rank == 0:
        print("Number of original instances:", n)
        print('Number of positive samples:', cnt)
        print("Done!")
