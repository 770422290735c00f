"""CPU restatement of the Jaccard pool-annotation pass (oracle; test infra only).

Follows ``retrieval_data_annotation.py`` (pure python sets + numpy, no torch).
"""
import numpy as np


def co_occurrence_ratio(seq_i, seq_j):
    """``retrieval_data_annotation.py:5-15``: ``|set(a)&set(b)| / |set(a)|set(b)|``;
    0 when either list is empty."""
    if not isinstance(seq_j, list):
        seq_j = [seq_j]
    if seq_i is None or seq_j is None:
        return 0
    if len(seq_i) == 0 or len(seq_j) == 0:
        return 0
    a, b = set(seq_i), set(seq_j)
    return len(a & b) / len(a | b)


def get_input_seq(seq):
    """``retrieval_data_annotation.py:17-20``: tokens between ``<|history|>`` and
    ``<|endofhistory|>`` (keeps the ego id and the ``<|timeK|>`` tokens)."""
    toks = seq.split('<|history|>')[1].split('<|endofhistory|>')[0].split(' ')
    return [t for t in toks if t != '']


def get_output_seq(seq):
    """``retrieval_data_annotation.py:22-26``: tokens between ``<|pre|>`` and
    ``<|endofpre|>`` minus anything containing ``'time'``."""
    toks = seq.split('<|pre|>')[1].split('<|endofpre|>')[0].split(' ')
    return [t for t in toks if t != '' and 'time' not in t]


def get_inout_list(data, gt):
    """``retrieval_data_annotation.py:28-34``."""
    return [get_input_seq(d) for d in data], [get_output_seq(g) for g in gt]


def occurrence_matrix(target, source):
    """``retrieval_data_annotation.py:36-41``: dense f64 [len(target), len(source)]."""
    m = np.zeros((len(target), len(source)))
    src_sets = [set(s) for s in source]           # same values as set(seq_j) per pair
    for i, t in enumerate(target):
        if len(t) == 0:
            continue
        ts = set(t)
        for j, ss in enumerate(src_sets):
            if len(ss) == 0:
                continue
            m[i, j] = len(ts & ss) / len(ts | ss)
    return m


def occurrence_matrix_naive(target, source):
    """Literal double loop over ``co_occurrence_ratio`` (the timed single-thread CPU
    baseline: two ``set()`` builds per pair, like the reference)."""
    m = np.zeros((len(target), len(source)))
    for i in range(len(target)):
        for j in range(len(source)):
            m[i, j] = co_occurrence_ratio(target[i], source[j])
    return m


def read_lines(path):
    """Blank-line filter of ``retrieval_data_annotation.py:140-159``."""
    with open(path, 'r') as f:
        return [line for line in f.read().splitlines() if (len(line) > 0 and not line.isspace())]


def rank_rows(m, stable=True):
    """``save_index_score`` ranking -- ``retrieval_data_annotation.py:88-89``
    (canonical tie-break: ascending index == stable argsort of -m)."""
    return np.argsort(-m, axis=1, kind="stable" if stable else None)


def train_positives(scores_out, threshold, dialog=False):
    """Positive lists of ``save_train_annotation`` -- ``retrieval_data_annotation.py:54,73-74``
    (strict ``>``; dialog keeps the first 4)."""
    out = []
    for i in range(scores_out.shape[0]):
        pos = np.where(scores_out[i] > threshold)[0].tolist()
        out.append(pos[:4] if dialog else pos)
    return out


def train_negative_candidates(scores_out, scores_in, threshold, neg_num=5):
    """Hard-negative candidate lists of ``save_train_annotation`` --
    ``retrieval_data_annotation.py:55-71`` with the candidate order taken from a
    STABLE argsort of ``-scores_in[i]`` (the reference order is tie-dependent,
    SURVEY.md section 8a quirk 9)."""
    res = []
    for i in range(scores_out.shape[0]):
        pos = set(np.where(scores_out[i] > threshold)[0].tolist())
        if not pos:
            res.append([])
            continue
        order = np.argsort(-scores_in[i], kind="stable")
        negs = []
        for idx in order:
            if idx not in pos and scores_out[i, idx] > 0:
                negs.append(int(idx))
            if len(negs) == neg_num:
                break
        if len(negs) < neg_num:
            for idx in order:
                if idx not in pos and scores_out[i, idx] == 0:
                    negs.append(int(idx))
                if len(negs) == neg_num:
                    break
        res.append(negs)
    return res


def sets_to_csr(seqs, vocab=None):
    """Token-string lists -> (indptr int32, sorted-unique ids int32, vocab dict).

    Any injective string->id map preserves the Jaccard ratio; ids are assigned
    densely in first-seen order unless ``vocab`` already holds the token.
    """
    vocab = {} if vocab is None else vocab
    indptr = [0]
    ids = []
    for s in seqs:
        row = sorted({vocab.setdefault(t, len(vocab)) for t in s})
        ids.extend(row)
        indptr.append(len(ids))
    return np.asarray(indptr, dtype=np.int32), np.asarray(ids, dtype=np.int32), vocab


def jaccard_csr(indptr_a, ids_a, indptr_b, ids_b, zero_diag=False):
    """Integer restatement on CSR sets: returns (intersection, union) int32 matrices
    and the f64 ratio (0 where either side is empty)."""
    na, nb = len(indptr_a) - 1, len(indptr_b) - 1
    nv = int(max(ids_a.max(initial=-1), ids_b.max(initial=-1))) + 1
    A = np.zeros((na, nv), dtype=np.int32)
    B = np.zeros((nb, nv), dtype=np.int32)
    for i in range(na):
        A[i, ids_a[indptr_a[i]:indptr_a[i + 1]]] = 1
    for j in range(nb):
        B[j, ids_b[indptr_b[j]:indptr_b[j + 1]]] = 1
    inter = A @ B.T
    la = np.diff(indptr_a)[:, None]
    lb = np.diff(indptr_b)[None, :]
    union = la + lb - inter
    ratio = np.zeros((na, nb))
    ok = (la > 0) & (lb > 0)
    np.divide(inter, union, out=ratio, where=ok)
    if zero_diag:
        np.fill_diagonal(ratio, 0)
    return inter, union, ratio
