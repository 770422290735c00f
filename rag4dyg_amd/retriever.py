"""Drop-in for the retriever's evaluation entry ``test()`` (``train/train_retriever.py:376-524``).

Same signature, same files (``resources/retrieval_result/<ds>/{val,test}_{index,score}.gen`` + results CSV),
same metric definitions; the arithmetic (encode, normalise, cosine scan, ranking) runs in the HIP library.
Differences, all documented in DESIGN.md:
  * ranking uses the canonical stable order (score desc, pool index asc) where the reference's
    ``np.argsort(-x)`` is unstable (SURVEY.md section 7 hard-part 1);
  * the pool is normalised once and stays in HBM (the reference redoes it per query batch, :435-436);
  * the lm_head GEMM whose result the reference discards (:419) is skipped;
  * ``args.rank_output``: "full" (default, reference-compatible full permutation rows) or "topk" (first
    ``args.topK`` indices per row -- what the generator actually consumes, ``dataloader/generator.py:46-48``).
"""
import os

import numpy as np
import torch
from torch.utils.data import DataLoader

from . import ops
from .dataloader import LineByLineTextDatasetHistory, get_dataloader, load_and_cache_examples
from .retrieval import PoolIndex, encode_batches
from .annotation import PHASES, phase            # R4D_PHASE_TIMING=1: wall-clock breakdown (tools/annotation_e2e.py)


def hit_rate_at_k(predictions, targets, k=1):
    """``train/train_retriever.py:31-38``."""
    gt = set(int(t) for t in targets)
    for i in set(int(p) for p in predictions[:k]):
        if i in gt:
            return 1
    return 0


def save_index_score(index_rows, score_matrix, save_index_file, save_score_file, steps):
    """Text layout of ``train/train_retriever.py:357-368`` (truncate at step 0, append afterwards, ``%.4f`` scores),
    plus a BINARY SIDE-CAR per file (``<file>.bin``: raw little-endian rows, int32 indices / float32 scores, and
    ``<file>.bin.json`` with dtype and row width) -- the text matrices are O(queries x pool) characters, the side-car is
    what ``read_matrix_rows`` (and through it this build's generator stage) loads when it is present and complete."""
    import json
    mode = 'w' if steps == 0 else 'a'
    with phase("text: format + write (*.gen, file-compatible)"):
        with open(save_index_file, mode) as f, open(save_score_file, mode) as g:
            for ind_row, sc_row in zip(np.asarray(index_rows).tolist(), np.asarray(score_matrix, dtype=np.float64).tolist()):
                f.write(' '.join(map(str, ind_row)) + '\n')
                g.write(' '.join([f"{x:.4f}" for x in sc_row]) + '\n')
    for path, arr, dt in ((save_index_file, index_rows, np.int32), (save_score_file, score_matrix, np.float32)):
        a = np.ascontiguousarray(arr, dtype=dt)
        meta_path = path + ".bin.json"
        meta = {"dtype": np.dtype(dt).name, "cols": int(a.shape[1]), "rows": 0}
        if steps != 0 and os.path.exists(meta_path):
            meta = json.load(open(meta_path))
            if meta["cols"] != a.shape[1] or meta["dtype"] != np.dtype(dt).name:
                raise ValueError(f"{path}.bin: row layout changed between steps")
        with phase("binary side-car write"):
            with open(path + ".bin", 'wb' if steps == 0 else 'ab') as h:
                h.write(a.tobytes())
        meta["rows"] += int(a.shape[0])
        with open(meta_path, "w") as h:
            json.dump(meta, h)


def read_matrix_rows(path, dtype):
    """Rows of a ``*_index`` / ``*_score`` matrix as a list of lists: from the binary side-car when it exists, is at
    least as new as the text file and has the size its header states; otherwise from the reference's text format."""
    import json
    b, m = path + ".bin", path + ".bin.json"
    if os.path.exists(b) and os.path.exists(m) and os.path.getmtime(b) >= os.path.getmtime(path) - 1e-3:
        meta = json.load(open(m))
        a = np.fromfile(b, dtype=np.dtype(meta["dtype"]))
        if a.size == meta["rows"] * meta["cols"]:
            return a.reshape(meta["rows"], meta["cols"]).astype(dtype).tolist()
    with open(path, encoding="utf-8") as f:
        return [list(map(dtype, line.split())) for line in f.read().splitlines() if len(line) > 0 and not line.isspace()]


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


@torch.no_grad()
def test(epoch, args, model, tokenizer, evaluate=True, prefix=""):
    test_mode = False if evaluate else True
    import time as _time
    _t_host = _time.perf_counter()
    eval_dataset = load_and_cache_examples(args, tokenizer, evaluate=evaluate, test=test_mode)
    if args.local_rank in [-1, 0]:
        os.makedirs(args.output_dir, exist_ok=True)
    eval_dataloader, args = get_dataloader(eval_dataset, tokenizer, args, split='eval')
    train_dataset = LineByLineTextDatasetHistory(tokenizer, args, file_path=args.train_data_file,
                                                 block_size=args.block_size)
    train_dataloader, args = get_dataloader(train_dataset, tokenizer, args, split='eval')
    model = _unwrap(model)
    print("Num examples = {}".format(len(eval_dataset)))
    print("Batch size = {}".format(args.eval_batch_size))
    hit_1, hit_3 = 0.0, 0.0
    nb_eval_steps = 0
    eval_loss = 0
    model.eval()
    device = args.device
    rank_output = getattr(args, "rank_output", "full")

    score_file = args.eval_data_gt_file if evaluate else args.test_data_gt_file
    with open(score_file, encoding="utf-8") as f:
        rows = [line for line in f.read().splitlines() if (len(line) > 0 and not line.isspace())]
    scores = torch.Tensor([list(map(float, item.split())) for item in rows])      # float32, as the reference (:409-410)
    scores = DataLoader(scores, batch_size=args.eval_batch_size, shuffle=False, num_workers=0, drop_last=False)
    if os.environ.get("R4D_PHASE_TIMING") == "1":
        PHASES["host: tokenise pool + queries, read the ground-truth score rows"] = \
            PHASES.get("host: tokenise pool + queries, read the ground-truth score rows", 0.0) + _time.perf_counter() - _t_host

    # HOT LOOP 1 (:414-422): pool embeddings, then one resident normalised index
    import torch.distributed as tdist
    world = tdist.get_world_size() if tdist.is_available() and tdist.is_initialized() else 1
    rank0 = world == 1 or tdist.get_rank() == 0
    if world > 1:        # one process per GPU: the pool ENCODE (the dominant cost) is sharded by whole batches, one all-gather
        from .dist import encode_pool_sharded
        train_embeddings = encode_pool_sharded(lambda bs: encode_batches(model, [b.to(device) for b in bs]),
                                               list(train_dataloader))
    else:
        with phase("device: upload + encode the pool"):
            train_embeddings = encode_batches(model, [batch.to(device) for batch in train_dataloader])    # fused groups
    print('size of train_embeddings: ', train_embeddings.size())
    index = PoolIndex(train_embeddings)
    n_pool = len(index)
    topk = min(max(3, int(getattr(args, "topK", 5))), n_pool)

    save_file_path = f'resources/retrieval_result/{args.dataset}/'
    os.makedirs(save_file_path, exist_ok=True)
    stem = 'val' if evaluate else 'test'
    save_index_file = os.path.join(save_file_path, f'{stem}_index.gen')
    save_score_file = os.path.join(save_file_path, f'{stem}_score.gen')

    # HOT LOOP 2 (:425-474).  The query batches are encoded up front in fused groups like the pool (every batch keeps
    # its own padding, so each row equals ``model.encode_meanpool(batch)`` of the reference's per-batch call).
    with phase("device: upload + encode the queries"):
        eval_batches = [batch.to(device) for batch in eval_dataloader]
        query_embeddings = encode_batches(model, eval_batches)
    row = 0
    for batch, score in zip(eval_batches, scores):
        h_egos = query_embeddings[row:row + batch.shape[0]]
        row += batch.shape[0]
        with phase("device: scan + top-k + BCE metric"):
            score = score.to(device)
            _vals, top_idx, dot_products = index.search(h_egos, topk, want_scores=True)
            loss = torch.nn.functional.binary_cross_entropy_with_logits(dot_products, score)      # metric only (:439-441)
            eval_loss += loss
        if prefix == "best" and rank0:
            if rank_output == "full":
                with phase("device: full-row argsort (file-compatible ranking)"):
                    perm = ops.argsort_desc(dot_products)
                with phase("download"):
                    index_rows = perm.cpu().numpy()
            else:
                index_rows = top_idx.cpu().numpy()
            with phase("download"):
                dp_host = dot_products.cpu().numpy()
            save_index_score(index_rows, dp_host, save_index_file, save_score_file, nb_eval_steps)
        # hit@1 / hit@3 against the top-3 of the float32 Jaccard rows (:458-474), canonical tie-break on both sides
        _, gt3 = ops.topk_f32(score.contiguous(), min(3, n_pool))
        gt3, pred = gt3.cpu().numpy(), top_idx.cpu().numpy()
        n = score.shape[0]
        hit_batch_1 = sum(hit_rate_at_k(pred[i], gt3[i], 1) for i in range(n))
        hit_batch_3 = sum(hit_rate_at_k(pred[i], gt3[i], 3) for i in range(n))
        hit_1 += hit_batch_1 / n
        hit_3 += hit_batch_3 / n
        nb_eval_steps += 1

    eval_loss = eval_loss / len(eval_dataset)                   # as the reference (:477)
    hit_1 = round(hit_1 / nb_eval_steps, 4)
    hit_3 = round(hit_3 / nb_eval_steps, 4)
    eval_metrics = {'hit@1': hit_1, 'hit@3': hit_3}

    result_save_file = os.path.join(save_file_path, "val_results.csv" if evaluate else "test_results.csv")
    if prefix == "best" and rank0:
        with open(result_save_file, "w") as f:                  # :485-495
            f.write(f"{'epoch'}, ")
            for param in args.para_names:
                f.write(f"{param}, ")
            f.write("Hit@1, Hit@3\n")
            f.write(f"{epoch}, ")
            for param in args.para_values:
                f.write(f"{param}, ")
            f.write(f"{hit_1},{hit_3}\n")
            f.write('\n')
    if evaluate:
        return eval_metrics, eval_loss
    if prefix == "best" and rank0:                              # :499-515 accumulate across runs
        save_folder = 'topk_scores_seed_retrieval' if getattr(args, "run_seed", False) else 'topk_scores_finetune'
        os.makedirs(save_folder, exist_ok=True)
        result_save_test = os.path.join(save_folder, args.dataset + '_retrieval.csv')
        with open(result_save_file) as f:
            lines = [ln for ln in f.read().splitlines() if ln.strip()]
        with open(result_save_test, 'a' if os.path.exists(result_save_test) else 'w') as g:
            g.write('\n'.join(lines[1:] if os.path.getsize(result_save_test) > 0 else lines) + '\n')
    return eval_metrics
