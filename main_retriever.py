#!/usr/bin/env python3
"""Drop-in for the reference ``main_retriever.py`` on the MI355X: same flags (``utils/args_parser_retriever.py``),
same checkpoint / tokenizer / result-file layout, training (``--do_train``: ``rag4dyg_amd.training.train``) and evaluation
(``--do_eval``) running on the gfx950 library.

Differences from the reference ``main()`` (``main_retriever.py:45-164``):
  * ``--do_train``: dropout masks come from the library's counter-based generator (keyed by ``--seed``), ``--fp16`` (apex)
    raises, resuming from ``--model_name_or_path checkpoint-<n>`` is not built (upstream restores the optimizer state and
    the step counters there, not the weights); started once per GPU the triples are sharded over the ranks and the
    gradients averaged (RCCL);
  * wandb is not imported (logging only); ``--model_name_or_path gpt2`` does not touch the network;
  * the eval batch stays ``per_gpu_eval_batch_size`` (32): ``n_gpu`` is pinned to 1 for BATCHING because a mean-
    pooled embedding depends on its padded batch (``train_retriever.py:420``) -- with 8 visible GPUs the reference
    itself would batch 256 and produce different embeddings; multi-GPU here means pool sharding instead: started
    once per GPU, the pool encode is split over the ranks by whole batches and all-gathered.
"""
import glob
import os
import random

import numpy as np
import torch
import torch.distributed

from rag4dyg_amd.cli_args import RETRIEVER, parse
from rag4dyg_amd.gpt2 import GPT2Config, GPT2LMHeadModelRAG, GPT2Model
from rag4dyg_amd.retriever import test
from rag4dyg_amd.tokenizer import WordLevelTokenizer, get_model_tokenizer

WEIGHTS_NAME = "pytorch_model.bin"
MODEL_CLASSES = {"gpt2": (GPT2Config, GPT2LMHeadModelRAG, WordLevelTokenizer)}
SIMPLEDYG_CKPT = {   # main_retriever.py:101-115
    "UCI_13": "simpledyg_ckpt/UCI_13/12/{42}/gpt2/checkpoint-0", "hepth": "simpledyg_ckpt/hepth/11/{4}/gpt2/checkpoint-0",
    "dialog": "simpledyg_ckpt/dialog/15/{7}/gpt2/checkpoint-0", "wikiv2": "simpledyg_ckpt/wikiv2/15/{42}/gpt2/checkpoint-0",
    "enron": "output/enron/simpledyg_ckpt/16/{42}/gpt2/checkpoint-0",
    "reddit": "output/reddit/simpledyg_ckpt/11/{42}/gpt2/checkpoint-0",
}


def set_seed(args):
    """``utils/model.py:15-20``."""
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(args.seed)


def main(argv=None):
    args = parse(RETRIEVER, "main_retriever.py", argv)
    set_seed(args)
    if args.dataset == "UCI_13":
        args.weight_decay = 1e-3
    if args.eval_data_file is None and args.do_eval:
        raise ValueError("--eval_data_file should be specified when do_eval is true")
    if args.no_cuda or not torch.cuda.is_available():
        raise SystemExit("main_retriever: needs the MI355X (rag4dyg_amd has no CPU fallback)")
    # one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, e.g. by torch.distributed.run): the pool encode
    # is sharded over the ranks by whole reference batches (rag4dyg_amd.dist.encode_pool_sharded); "nccl" IS RCCL on ROCm
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", max(args.local_rank, 0))) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    args.device = torch.device("cuda", local)
    args.n_gpu = 1
    if world > 1 and not torch.distributed.is_initialized():
        torch.distributed.init_process_group(backend=os.environ.get("R4D_DIST_BACKEND", "nccl"))
        args.local_rank = -1                           # every rank evaluates; rank 0 writes the files
    lr_type = 'y' if args.learning_rate > 0 else 'n'
    ckpt = 1 if args.should_continue == 1 else 0
    args.para_names = ['d', 'alpha', 'eta', 'gamma', 'nl', 'nh', 'emb', 'bz', 'lr', 'lrdecay', 'tdecay', 'se', 'temp',
                       'ckpt', 'wd', 'loss']
    args.para_values = [args.dataset, args.alpha, args.eta, args.gamma, args.n_layer, args.n_head, args.n_embed,
                        args.per_gpu_train_batch_size, args.learning_rate, lr_type, args.lambda_decay, args.seed,
                        args.temperature, ckpt, args.weight_decay, args.loss_type]
    args.run_name = ''.join(f"{n}:{v}_" for n, v in zip(args.para_names, args.para_values))

    model, tokenizer, model_class, args = get_model_tokenizer(args, MODEL_CLASSES)
    if args.should_continue:
        print('load model from checkpoint')
        simpledyg_checkpoint = args.simpledyg_checkpoint or SIMPLEDYG_CKPT[args.dataset]
        model.transformer = GPT2Model.from_pretrained(simpledyg_checkpoint)
        model.resize_token_embeddings(len(tokenizer))
        model.tie_weights()
    model = model.to(args.device)

    if args.do_train:                                  # main_retriever.py:124-136 -> train/train_retriever.train
        from rag4dyg_amd.dataloader import load_and_cache_examples
        from rag4dyg_amd.training import train
        train_dataset = load_and_cache_examples(args, tokenizer, evaluate=False)
        global_step, train_loss = train(args, train_dataset, model, tokenizer)
        print(" global_step = {}, average loss = {}".format(global_step, train_loss))
    if args.do_eval and args.local_rank in [-1, 0]:
        checkpoints = [args.output_dir]
        if args.eval_all_checkpoints:
            checkpoints = list(os.path.dirname(c) for c in
                               sorted(glob.glob(args.output_dir + "/**/" + WEIGHTS_NAME, recursive=True)))
        print("Evaluate the following checkpoints: {}".format(checkpoints))
        from rag4dyg_amd.annotation import phase       # R4D_PHASE_TIMING=1: wall-clock breakdown (tools/annotation_e2e.py)
        for checkpoint in checkpoints:
            with phase("host: read checkpoint + load_state_dict + upload"):
                state_dict = torch.load(os.path.join(checkpoint, WEIGHTS_NAME), map_location="cpu", weights_only=True)
                model.load_state_dict(state_dict)      # strict, like main_retriever.py:152-153; no re-tie (as upstream)
                model.to(args.device)
            test_metrics = test(0, args, model, tokenizer, evaluate=False, prefix="best")
            print('test_metrics: ', test_metrics)


if __name__ == "__main__":
    main()
