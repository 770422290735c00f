"""SimpleDyG greedy link-prediction evaluation, forward passes on the gfx950 kernels.

Mirrors ``utils/Evaluation_SimpleDyG.py``: ``Evaluation`` (natural-log NDCG@k, Jaccard, recall, precision, MAP) and
``get_eval_metrics`` -- greedy decode per test sequence, stop on ``<|endoftext|>`` (val: after 10 tokens; test: at
``n_ctx - len(spl_tokens)``), then the predicted node list is scored against the ground truth.  ``greedy_decode`` is the
reference's loop verbatim (batch 1, FULL forward on the growing sequence each step, no KV cache, :126-134);
``greedy_decode_batch`` -- what ``get_eval_metrics`` runs -- decodes ``--per_gpu_eval_batch_size`` sequences together with
the key/value cache (``r4d_gpt2_decode_step_f32``).  Only the last position's logits are needed, so the tied lm_head
GEMM runs on one row per sequence and step.
Reference quirk kept: ``get_eval_metrics`` always reads ``args.eval_data_file`` / ``eval_data_gt_file``, also in
mode="test" (:57-58).
"""
import json
import math
import os

import numpy as np
import torch

from . import ops


class Evaluation:
    """``utils/Evaluation_SimpleDyG.py:14-51``."""

    def jaccard(self, pred, label):
        pred, label = set(pred), set(label)
        return len(pred & label) / len(pred | label)

    def ndcg_k(self, sorted_indices, ground_truth, k):
        dcg, pdcg = 0, 0
        for i, item in enumerate(sorted_indices[:k]):
            if item in ground_truth:
                dcg += 1 / math.log(i + 2)
        for i in range(min(len(ground_truth), k)):
            pdcg += 1 / math.log(i + 2)
        return dcg / pdcg

    def map_k(self, sort, y, k):
        sum_precs, hists = 0, 0
        for n, item in enumerate(sort[:k]):
            if item in y:
                hists += 1
                sum_precs += hists / (n + 1)
        return sum_precs

    def recall_k(self, sort, y, k):
        return sum(1 for y_i in y if y_i in sort[:k]) / len(y)

    def precision_k(self, sort, y, k):
        return sum(1 for y_i in y if y_i in sort[:k]) / k


@torch.no_grad()
def greedy_next_token(model, indexed_tokens, device):
    """argmax of the last position's logits for one sequence (``outputs[0][0, -1, :]``, :127-129)."""
    ids = torch.tensor([indexed_tokens], dtype=torch.int64, device=device)
    hidden = model.transformer.encode(ids, want_hidden=True)["hidden"]
    logits = ops.lm_logits(hidden[:, -1, :].contiguous(), model.lm_head.weight)      # == wte when tied
    return int(torch.argmax(logits[0]).item())


@torch.no_grad()
def greedy_decode(model, tokenizer, indexed_tokens, mode, max_len, n_spl, device):
    """The decode loop of ``get_eval_metrics`` (:120-145) for one sequence; returns the extended id list."""
    indexed_tokens = list(indexed_tokens)
    eos = tokenizer.encode("<|endoftext|>")
    gen_len = 0
    while True:
        predicted_index = greedy_next_token(model, indexed_tokens, device)
        indexed_tokens.append(predicted_index)
        gen_len += 1
        if mode == "val":
            if gen_len > 10:
                break
        elif len(indexed_tokens) >= max_len - n_spl:
            break
        if predicted_index in eos:                     # == decode(...).endswith('<|endoftext|>') / the while condition
            break
    return indexed_tokens


@torch.no_grad()
def greedy_decode_batch(model, tokenizer, token_lists, mode, max_len, n_spl, device):
    """``greedy_decode`` for many sequences at once with the key/value cache (``GPT2Model.prefill`` / ``decode_step``):
    the prompts are prefilled in a few length-grouped forwards (``prefill_last``), then every token is one cached step
    over all sequences, argmax and stop rules on the device (``GreedyDecoder``).  Same ids as the one-at-a-time loop up to fp32 summation order (the reference re-runs the full
    forward per token, ``Evaluation_SimpleDyG.py:126-134``)."""
    tr = model.transformer
    n = len(token_lists)
    if n == 0:
        return []
    eos = tokenizer.encode("<|endoftext|>")
    toks = [list(t) for t in token_lists]
    tmax = max(len(t) for t in toks)
    budget = 11 if mode == "val" else max(1, max_len - n_spl - min(len(t) for t in toks))
    cap = min(tmax + budget + 1, tr.wpe.num_embeddings)
    ids_h = np.zeros((n, tmax), dtype=np.int64)                        # built on the host: ONE upload
    for i, t in enumerate(toks):
        ids_h[i, :len(t)] = t
    ids = torch.from_numpy(ids_h).to(device)
    dec = tr.greedy_decoder(n, cap)
    last = tr.prefill_last(dec.cache, [len(t) for t in toks], input_ids=ids)
    lens = torch.tensor([len(t) for t in toks], dtype=torch.int32, device=device)
    # the reference's stop rules (Evaluation_SimpleDyG.py:135-145) evaluated on the device after every token:
    # val: 11 tokens; test: <|endoftext|> or len(tokens) >= n_ctx - len(spl_tokens); always: the cache is full
    limit = cap if mode == "val" else min(cap, max_len - n_spl)
    gen = dec.run(last, lens, 11 if mode == "val" else cap, limit, eos)
    for t, g in zip(toks, gen):
        t.extend(g)
    return toks


def get_eval_metrics(args, model, tokenizer, step, mode="val"):
    """Drop-in for ``utils/Evaluation_SimpleDyG.get_eval_metrics`` (:53-211): NDCG@5 / Jaccard over the file pair,
    results CSV + per-sample JSON under ``<output_dir>/results[_seed_jac]/<mode>_score``."""
    spl_tokens = tokenizer.additional_special_tokens + [tokenizer.bos_token, tokenizer.eos_token, tokenizer.pad_token]
    with open(args.eval_data_file, encoding="utf-8") as f:
        data = [line for line in f.read().splitlines() if (len(line) > 0 and not line.isspace())]
    with open(args.eval_data_gt_file, encoding="utf-8") as f:
        data_gt = [line for line in f.read().splitlines() if (len(line) > 0 and not line.isspace())]
    assert len(data) == len(data_gt)
    with open(os.path.join('./vocabs', args.dataset, str(args.timestamp), 'vocab.json')) as f:
        vocab = json.load(f)
    sub = "results_seed_jac" if getattr(args, "run_seed", False) else "results"
    save_score_path = os.path.join(args.output_dir, sub, mode + '_score')
    os.makedirs(save_score_path, exist_ok=True)

    Eval = Evaluation()
    model.eval()
    device = next(model.parameters()).device
    MAX_LEN = model.config.n_ctx
    topk = [5]
    metric_terms = ['MAP', 'NDCG', 'jaccard']
    top_k_scores = {metric: len(topk) * [0] for metric in metric_terms}
    generated_dict = {}
    num_user_test = 0
    jobs = []
    for i, (input_text, text_gt) in enumerate(zip(data, data_gt)):
        generated_dict[i] = {}
        user_id = input_text.split()[2]
        target_list = [t for t in text_gt.split()[1:-2] if t != user_id and t in vocab]
        if len(target_list) == 0:
            print('text_gt: ', text_gt)
            continue
        indexed_tokens = tokenizer.encode(input_text)
        num_user_test += 1
        if len(indexed_tokens) > MAX_LEN:
            print('len_input: ', len(indexed_tokens))
            indexed_tokens = indexed_tokens[-1000:]
        jobs.append((i, input_text, user_id, target_list, indexed_tokens, num_user_test))
    bs = max(1, int(getattr(args, "per_gpu_eval_batch_size", 32) or 32))
    # independent sequences: one cached decode step serves a whole batch; started once per GPU (torch.distributed
    # initialised by main_SimpleDyG.py) a rank decodes every world-th batch, the ids are all-gathered and EVERY rank
    # scores all sequences in file order -- the sums of the single-process run
    import torch.distributed as dist
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
    chunks = [jobs[b0:b0 + bs] for b0 in range(0, len(jobs), bs)]
    generated = {}
    for chunk in chunks[rank::world]:
        outs = greedy_decode_batch(model, tokenizer, [j[4] for j in chunk], mode, MAX_LEN, len(spl_tokens), device)
        for job, out_ids in zip(chunk, outs):
            generated[job[0]] = [int(t) for t in out_ids]
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, generated)
        generated = {k: v for part in parts for k, v in part.items()}
    for (i, input_text, user_id, target_list, indexed_tokens, nut) in jobs:
        out_ids = generated[i]
        predicted_list = tokenizer.decode(out_ids).split()[len(indexed_tokens):]
        predicted = [t for t in predicted_list if t != user_id and t not in spl_tokens]
        for topi, k in enumerate(topk):
            try:
                top_k_scores['NDCG'][topi] += Eval.ndcg_k(predicted, target_list, k)
            except ZeroDivisionError:
                pass
            top_k_scores['jaccard'][topi] += Eval.jaccard(predicted, target_list)
        generated_dict[i].update({'user_id': user_id, 'input': input_text, 'target_list': target_list,
                                  'len input_text': len(input_text.split()), 'predicted_list_ori': predicted_list,
                                  'predicted': predicted, 'NDCG@k': str(Eval.ndcg_k(predicted, target_list, 1)),
                                  'num_user_test': str(nut)})
    for metric in metric_terms:
        for topi, _k in enumerate(topk):
            top_k_scores[metric][topi] = round(top_k_scores[metric][topi] / max(num_user_test, 1), 4)
    if rank != 0:                                   # rank 0 writes the files
        return top_k_scores
    result_save_file = os.path.join(save_score_path, mode + '_results_epoch.csv')
    if not os.path.exists(result_save_file):
        with open(result_save_file, 'w') as f:
            f.write(''.join(p + ',' for p in args.para_names))
            f.write(''.join('NDCG@{},'.format(k) for k in topk) + ''.join('jaccard@{},'.format(k) for k in topk) + '\n')
    with open(result_save_file, 'a') as f:
        f.write(''.join(str(v) + ',' for v in args.para_values))
        f.write(''.join(str(top_k_scores['NDCG'][j]) + ',' for j in range(len(topk))))
        f.write(''.join(str(top_k_scores['jaccard'][j]) + ',' for j in range(len(topk))) + '\n')
    with open('{}.json'.format(save_score_path + '/eval_results_' + str(step)), 'wt') as f:
        json.dump(generated_dict, f, indent=4)
    return top_k_scores
