#!/usr/bin/env python3
"""GPU box, step 1 of the TRAINED-WEIGHT parity vectors (G10): train the retriever on the real UCI_13/12 data with this build's
own trainer (tools/train_uci13_demo.py's workdir, the reference script's hyper-parameters unless overridden), then run the
hot path -- encode the 110 test queries and the 1,708 pool histories, score, top-10 -- on the trained checkpoint and dump
weights + device outputs under gpurun_out/g10/<tag>.  Step 2 runs in the build container, where the reference is importable:
``python oracle/gen_golden.py g10 <tag> ...`` loads the SAME weights into the reference model on CPU and writes the fixture /
the comparison report.
    python tools/g10_trained.py <tag> <n_layer> <n_head> <n_embed> <epochs> <lr>"""
import glob
import io
import os
import sys
import tempfile
import time
from contextlib import redirect_stdout

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import train_uci13_demo as demo  # noqa: E402


def main():
    tag, L, H, d, epochs, lr = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    out_dir = os.path.join(REPO, "gpurun_out", "g10")
    os.makedirs(out_dir, exist_ok=True)
    root = tempfile.mkdtemp(prefix="r4d_g10_")
    base, ret = demo.build_workdir(root)
    os.chdir(root)
    import main_retriever
    out = os.path.join(root, "out")
    argv = (f"--dataset UCI_13 --timestamp 12 --eta 0.8 --gamma 0.4 --temperature 0.1 --alpha 1 --lambda_decay 0.0001 --lrdecay 1 "
            f"--warmup_steps 0 --output_dir {out} --model_type gpt2 --model_name_or_path gpt2 --train_data_file {base}/train.link_prediction "
            f"--train_pair_data_file {ret}/train_index.retrieval --do_train --eval_data_file {base}/val.link_prediction "
            f"--eval_data_gt_file {ret}/val_score.retrieval --test_data_file {base}/test.link_prediction "
            f"--test_data_gt_file {ret}/test_score.retrieval --per_gpu_train_batch_size 64 --num_train_epochs {epochs} "
            f"--learning_rate {lr} --n_layer {L} --n_head {H} --n_embed {d} --block_size 512 --seed 42 --patience 1000 --topK 5").split()
    t0 = time.time()
    buf = io.StringIO()
    with redirect_stdout(buf):
        main_retriever.main(argv)
    log = [ln for ln in buf.getvalue().splitlines() if ln.startswith("epoch ")]
    print(log[0] if log else "", "\n", log[-1] if log else "", f"\n# trained in {time.time() - t0:.0f} s", flush=True)
    ck = sorted(glob.glob(os.path.join(out, "**", "checkpoint-1"), recursive=True))[-1]            # the LAST epoch weights
    sd = torch.load(os.path.join(ck, "pytorch_model.bin"), map_location="cpu")
    sd = {k: v for k, v in sd.items() if not k.endswith(".attn.bias") and not k.endswith("masked_bias")}
    if "lm_head.weight" in sd and torch.equal(sd["lm_head.weight"], sd["transformer.wte.weight"]):
        del sd["lm_head.weight"]                                                  # tied: the reference re-ties on load
    sd["transformer.wpe.weight"] = sd["transformer.wpe.weight"][:512].clone()     # block_size 512: rows past it are never read
    np.savez(os.path.join(out_dir, f"{tag}_weights.npz"), **{k: v.numpy() for k, v in sd.items()})

    # the hot path on the trained checkpoint (device)
    from rag4dyg_amd import ops
    from rag4dyg_amd.gpt2 import GPT2LMHeadModelRAG
    from rag4dyg_amd.retrieval import PoolIndex, encode_batches, right_pad_batches
    dev = torch.device("cuda:0")
    model = GPT2LMHeadModelRAG.from_pretrained(ck).to(dev).eval()
    g6 = np.load(os.path.join(REPO, "tests", "golden", "g6_uci_tokens.npz"))
    pad = int(g6["pad_id"])

    def seqs(flat, off):
        return [flat[off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]
    pool, test = seqs(g6["pool_flat"], g6["pool_off"]), seqs(g6["test_flat"], g6["test_off"])
    res = {}
    for mode in ("split3", "f32"):
        ops.set_gemm_split3(mode == "split3")
        model.transformer.__dict__.pop("_w3_cache", None)
        pe = encode_batches(model, right_pad_batches(pool, 32, pad, dev))
        qe = encode_batches(model, right_pad_batches(test, 32, pad, dev))
        vals, idx, S = PoolIndex(pe).search(qe, 10, want_scores=True)
        res.update({f"{mode}_pool_emb_head": pe[:256].cpu().numpy(), f"{mode}_query_emb": qe.cpu().numpy(),
                    f"{mode}_scores": S.cpu().numpy(), f"{mode}_top10": idx.cpu().numpy()})
        res[f"{mode}_pool_emb_colsum"] = pe.double().sum(0).cpu().numpy()
    ops.set_gemm_split3(True)
    np.savez(os.path.join(out_dir, f"{tag}_device.npz"), **res)
    w = {k: v.float() for k, v in sd.items()}
    stats = {"max|w|": max(float(v.abs().max()) for v in w.values()),
             "ln gains": [round(float(w[k].min()), 3) for k in w if k.endswith("ln_1.weight")][:2] + [round(float(w[k].max()), 3) for k in w if k.endswith("ln_f.weight")],
             "wte std": round(float(w["transformer.wte.weight"].std()), 4)}
    print("weights:", stats, {k: os.path.getsize(os.path.join(out_dir, k)) for k in os.listdir(out_dir) if k.startswith(tag)})


if __name__ == "__main__":
    main()
