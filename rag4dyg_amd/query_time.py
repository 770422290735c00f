"""Query times of the training sequences -- the ``resources/<dataset>_train_query_time.pt`` file the retriever's training loop
reads (``train/train_retriever.py:289-291``; produced upstream by ``get_train_query_time.py``).

For the ego node of every training line: among its events in BOTH directions (``get_train_query_time.py:7-15``) up to snapshot
``timestamp - 2`` take the latest snapshot present; the query time is the time stamp of the node's last event BEFORE that
snapshot, or -- when all of its events lie in that one snapshot -- its last event inside it (``:17-25``); divided by the
dataset's time unit (``:47-54``) and stored as float32.  Host-side data preparation (numpy; no GPU work).
"""
import os

import numpy as np

from .dataloader import read_nonblank_lines

SCALES = {"UCI_13": 3600 * 24, "hepth": 3600 * 24 * 30, "dialog": 1, "wikiv2": 3600 * 24, "enron": 1, "reddit": 1}


def query_times(u, i, ts, snapshot, egos, timestamp, scale):
    """``u, i, ts, snapshot``: the columns of ``ml_<dataset>.csv`` (one row per undirected event); ``egos``: the ego id of every
    training line.  Returns float32 [len(egos)]."""
    u, i = np.asarray(u, np.int64), np.asarray(i, np.int64)
    ts, snapshot = np.asarray(ts, np.float64), np.asarray(snapshot, np.int64)
    node = np.concatenate([u, i])                                      # every event seen from both of its ends
    t2 = np.concatenate([ts, ts])
    s2 = np.concatenate([snapshot, snapshot])
    keep = s2 <= int(timestamp) - 2
    node, t2, s2 = node[keep], t2[keep], s2[keep]
    order = np.argsort(node, kind="stable")
    node, t2, s2 = node[order], t2[order], s2[order]
    starts = np.searchsorted(node, np.arange(int(node.max()) + 2 if node.size else 1))
    cache, out = {}, np.empty(len(egos), np.float64)
    for n, q in enumerate(egos):
        q = int(q)
        if q not in cache:
            if q < 0 or q + 1 >= len(starts) or starts[q] == starts[q + 1]:
                raise ValueError(f"query_times: node {q} has no event up to snapshot {int(timestamp) - 2}")
            tq, sq = t2[starts[q]:starts[q + 1]], s2[starts[q]:starts[q + 1]]
            before = tq[sq < sq.max()]
            cache[q] = float(before.max() if before.size else tq.max())
        out[n] = cache[q] / scale
    return out.astype(np.float32)


def main(argv=None):
    """``python get_train_query_time.py <dataset> <timestamp>`` (cwd-relative ``resources/`` like upstream)."""
    import sys
    import pandas as pd
    import torch
    argv = sys.argv if argv is None else argv
    data_name, timestamp = argv[1], argv[2]
    tab = pd.read_csv(os.path.join("resources", data_name, timestamp, f"ml_{data_name}.csv"))
    lines = read_nonblank_lines(os.path.join("resources", data_name, timestamp, "train.link_prediction"))
    egos = [int(line.split('<|history|>')[1].split(' ')[1]) for line in lines]
    times = query_times(tab["u"], tab["i"], tab["ts"], tab["timestamp"], egos, timestamp, SCALES[data_name])
    torch.save(torch.from_numpy(times), os.path.join("resources", data_name + "_train_query_time.pt"))
    print(f"{len(times)} query times -> resources/{data_name}_train_query_time.pt")
