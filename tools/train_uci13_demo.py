#!/usr/bin/env python3
"""Real-data end-to-end run of the retriever TRAINING path: ``main_retriever.py --do_train`` with the
hyper-parameters of ``scripts/train_retriever/train_retriever_UCI_13.sh`` on the shipped UCI_13/12 data -- rebuilt here from the
committed fixtures, because the reference tree does not exist on the GPU box: token ids of the 1,708 training / 146 validation /
110 test histories (G6), the Jaccard ground-truth rows and the reference's own annotation triples (G5), the query times (G9).
Prints one line per epoch (training loss, validation loss and hit@3 / hit@5 ... as the loop reports them) and the test metrics of
the best and the last weights.      python tools/train_uci13_demo.py [epochs] [learning rate]
"""
import io
import json
import os
import re
import sys
import tempfile
import time
from contextlib import redirect_stdout

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
GOLD = os.path.join(REPO, "tests", "golden")


def build_workdir(root):
    g6 = np.load(os.path.join(GOLD, "g6_uci_tokens.npz"))
    g5 = np.load(os.path.join(GOLD, "g5_jaccard_UCI_13.npz"))
    g9 = np.load(os.path.join(GOLD, "g9_query_times.npz"))
    v0, t = int(g6["vocab_size"]), 12
    names = {v0: "<|endoftext|>", v0 + 1: "<|history|>", v0 + 2: "<|endofhistory|>", v0 + 3: "<|pre|>", v0 + 4: "<|endofpre|>"}
    names.update({v0 + 5 + k: f"<|time{k}|>" for k in range(t + 1)})

    def lines(flat, off, tail=""):
        return [" ".join(names.get(int(x), str(int(x))) for x in flat[off[i]:off[i + 1]]) + tail for i in range(len(off) - 1)]
    base = os.path.join(root, "resources", "UCI_13", "12")
    ret = os.path.join(base, "train_retrieval")
    os.makedirs(ret)
    os.makedirs(os.path.join(root, "vocabs", "UCI_13", "12"))
    json.dump({str(i): i for i in range(v0)}, open(os.path.join(root, "vocabs", "UCI_13", "12", "vocab.json"), "w"))
    # the loaders read the history part only; the prediction part of a training line is a placeholder
    open(os.path.join(base, "train.link_prediction"), "w").write(
        "\n".join(lines(g6["pool_flat"], g6["pool_off"], " <|pre|> <|time12|> 0 <|endofpre|> <|endoftext|>")) + "\n")
    open(os.path.join(base, "val.link_prediction"), "w").write("\n".join(lines(g6["val_flat"], g6["val_off"])) + "\n")
    open(os.path.join(base, "test.link_prediction"), "w").write("\n".join(lines(g6["test_flat"], g6["test_off"])) + "\n")
    for name, m in (("val", g5["m_val"]), ("test", g5["m_test"])):
        with open(os.path.join(ret, f"{name}_score.retrieval"), "w") as f:
            for row in m:
                f.write(" ".join(str(x) for x in row) + "\n")
    with open(os.path.join(ret, "train_index.retrieval"), "w") as f:
        for a, p, n in g5["ann_triples"]:
            f.write(f"{int(a)} {int(p)} {int(n)}\n")
    torch.save(torch.from_numpy(g9["UCI_13_times"]), os.path.join(root, "resources", "UCI_13_train_query_time.pt"))
    return base, ret


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    lr = sys.argv[2] if len(sys.argv) > 2 else "1e-5"
    root = tempfile.mkdtemp(prefix="r4d_uci13_")
    base, ret = build_workdir(root)
    os.chdir(root)
    import main_retriever
    out = os.path.join(root, "out")
    argv = (f"--dataset UCI_13 --timestamp 12 --eta 0.8 --gamma 0.4 --temperature 0.1 --alpha 1 --lambda_decay 0.0001 --lrdecay 1 "
            f"--warmup_steps 0 --output_dir {out} --model_type gpt2 --model_name_or_path gpt2 --train_data_file {base}/train.link_prediction "
            f"--train_pair_data_file {ret}/train_index.retrieval --do_train --eval_data_file {base}/val.link_prediction "
            f"--eval_data_gt_file {ret}/val_score.retrieval --test_data_file {base}/test.link_prediction "
            f"--test_data_gt_file {ret}/test_score.retrieval --per_gpu_train_batch_size 64 --num_train_epochs {epochs} "
            f"--learning_rate {lr} --n_layer 4 --n_head 2 --n_embed 512 --block_size 512 --seed 42 --patience 50 --topK 5").split()
    print(f"# main_retriever.py {' '.join(a for a in argv if not a.startswith(root))[:0]}--do_train on UCI_13/12 (1,708 training histories, "
          f"9,578 annotation triples, batch 64, lr {lr}, {epochs} epochs, L4 H2 d512, dropout 0.1, seed 42)", flush=True)
    buf = io.StringIO()
    t0 = time.time()
    with redirect_stdout(buf):
        main_retriever.main(argv)
    log = buf.getvalue()
    for line in log.splitlines():
        if re.match(r"epoch \d+:", line) or line.startswith("test_metrics"):
            print(line)
    print(f"# wall time {time.time() - t0:.1f} s ({epochs} epochs incl. validation after each and four final evaluation passes)")


if __name__ == "__main__":
    main()
