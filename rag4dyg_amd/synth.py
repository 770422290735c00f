"""Synthetic temporal-neighbour sequences of the reference's shapes.

Token-id layout (verified against tokenizers/UCI_13/12/tokenizer.json): node ids 0..V0-1, then
<|endoftext|>=V0, <|history|>=V0+1, <|endofhistory|>=V0+2, <|pre|>=V0+3, <|endofpre|>=V0+4,
<|time0..t|>=V0+5..V0+5+t, [PAD]=V0+6+t, [MASK]=V0+7+t.

Sequence grammar (csv2resources.py:125-164): <|endoftext|> <|history|> ego <|time a|> n.. <|time a+1|> n.. <|endofhistory|>
Lengths follow a log-normal fitted to the measured histograms; node ids are Zipf(1.1) over V0 (hub-heavy).
"""
from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class Shape:
    name: str
    v0: int            # node vocabulary
    t: int             # --timestamp (time tokens 0..t)
    n_layer: int
    n_head: int
    n_embd: int
    pool_len: tuple    # (median, p99, cap) of pool (history) lengths
    query_len: tuple   # (median, p99, cap) of query lengths
    block_size: int = 512

    @property
    def vocab(self):   # len(tokenizer), retriever flavour (with [MASK]; utils/tokenizer.py:44-48)
        return self.v0 + 8 + self.t

    @property
    def pad_id(self):
        return self.v0 + 6 + self.t

    @property
    def vocab_generator(self):   # len(tokenizer), SimpleDyG / generator flavour (no [MASK]; main_SimpleDyG.py:91-95)
        return self.v0 + 7 + self.t


# architecture: scripts/train_retriever/train_retriever_{UCI_13,wikiv2}.sh; lengths: SURVEY.md section 5 / 8d
UCI_13 = Shape("UCI_13", 1781, 12, 4, 2, 512, (13, 158, 329), (59, 304, 339))
WIKIV2 = Shape("wikiv2", 8794, 15, 2, 6, 768, (6, 86, 512), (43, 225, 512))
HEPTH = Shape("hepth", 4737, 11, 12, 2, 256, (9, 39, 78), (9, 39, 78), 1024)
# generator shape of BASELINE config 5: scripts/train_generator/train_rag_graphpooling_reddit_seed.sh:6-10 (L2 H8 d512),
# V0 = 11,901 (len(tokenizer) = 11,919 without [MASK]), pool 10,527, history length p50/p99 = 8/133
REDDIT = Shape("reddit", 11901, 11, 2, 8, 512, (8, 133, 512), (8, 133, 512))
SHAPES = {"UCI_13": UCI_13, "wikiv2": WIKIV2, "hepth": HEPTH, "reddit": REDDIT}


def _lengths(rng, n, median, p99, cap, lo=6):
    sigma = (np.log(p99) - np.log(median)) / 2.3263
    ln = np.exp(rng.normal(np.log(median), sigma, size=n))
    return np.clip(np.rint(ln), lo, cap).astype(np.int64)


def _zipf_nodes(rng, n, v0, a=1.1):
    # truncated Zipf over ranks 1..v0 mapped through a fixed permutation (hubs are arbitrary ids)
    w = 1.0 / np.arange(1, v0 + 1) ** a
    w /= w.sum()
    return rng.choice(v0, size=n, p=w)


def sequences(shape, n, kind="pool", seed=2026):
    """n ragged token-id sequences (list of int64 arrays) following the history grammar."""
    rng = np.random.default_rng(seed)
    med, p99, cap = shape.pool_len if kind == "pool" else shape.query_len
    lens = _lengths(rng, n, med, p99, min(cap, shape.block_size))
    perm = np.random.default_rng(7).permutation(shape.v0)
    eot, hist, endhist, time0 = shape.v0, shape.v0 + 1, shape.v0 + 2, shape.v0 + 5
    out = []
    for L in lens:
        L = int(L)
        nslots = int(rng.integers(1, shape.t + 1))
        nslots = min(nslots, max(1, L - 4))
        body = L - 4 - nslots                      # eot, hist, ego, endhist + time tokens
        cuts = np.sort(rng.integers(0, body + 1, size=nslots - 1)) if nslots > 1 else np.zeros(0, np.int64)
        sizes = np.diff(np.concatenate([[0], cuts, [body]])).astype(np.int64)
        start = int(rng.integers(0, shape.t + 1 - nslots + 1))
        nodes = perm[_zipf_nodes(rng, max(body, 0) + 1, shape.v0)]
        seq = [eot, hist, int(nodes[0])]
        p = 1
        for s_i, sz in enumerate(sizes):
            seq.append(time0 + start + s_i)
            seq.extend(int(x) for x in nodes[p:p + sz])
            p += int(sz)
        seq.append(endhist)
        out.append(np.asarray(seq, dtype=np.int64))
    return out


def output_sets(shape, n, seed=2026, in_sets=False):
    """Synthetic Jaccard sets (CSR ptr/idx int32): out-sets ~ 1+Geom(0.55) capped 72, in-sets ~ lognormal mean 18 cap 238."""
    rng = np.random.default_rng(seed + (1 if in_sets else 0))
    if in_sets:
        sizes = np.clip(np.rint(np.exp(rng.normal(np.log(14), 0.7, size=n))), 2, 238).astype(np.int64)
    else:
        sizes = np.minimum(1 + rng.geometric(0.55, size=n) - 1, 72).astype(np.int64)
    perm = np.random.default_rng(7).permutation(shape.v0)
    ptr = np.zeros(n + 1, np.int32)
    idx = []
    for i, s in enumerate(sizes):
        row = np.unique(perm[_zipf_nodes(rng, int(s), shape.v0)])
        idx.append(row)
        ptr[i + 1] = ptr[i] + len(row)
    return ptr, np.concatenate(idx).astype(np.int32)
