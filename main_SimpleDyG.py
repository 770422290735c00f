#!/usr/bin/env python3
"""Drop-in for the reference ``main_SimpleDyG.py`` (flags of ``utils/args_parser_SimpleDyG.py``): the SimpleDyG
GPT-2 FORWARD on the MI355X -- ``--do_eval`` runs the reference's greedy link-prediction evaluation
(``get_eval_metrics``, ``utils/Evaluation_SimpleDyG.py:53-211``: NDCG@5 / Jaccard, as ``main_SimpleDyG.py:485-487``) for
every checkpoint and also reports the LM loss over ``--eval_data_file`` (``evaluate``, :345-372).  LM training
(``train``/``train_epoch`` :148-343, backward pass) is not part of this build and raises."""
import glob
import os

import torch
import torch.distributed

from rag4dyg_amd.cli_args import SIMPLEDYG, parse
from rag4dyg_amd.dataloader import LineByLineTextDataset, get_dataloader
from rag4dyg_amd.evaluation import get_eval_metrics
from rag4dyg_amd.gpt2 import GPT2Config, GPT2LMHeadModel
from rag4dyg_amd.tokenizer import WordLevelTokenizer, get_model_tokenizer

WEIGHTS_NAME = "pytorch_model.bin"
MODEL_CLASSES = {"gpt2": (GPT2Config, GPT2LMHeadModel, WordLevelTokenizer)}


@torch.no_grad()
def evaluate(args, model, tokenizer, prefix=""):
    """``main_SimpleDyG.py:345-372``: mean over batches of the shifted-CE LM loss, labels == inputs."""
    eval_dataset = LineByLineTextDataset(tokenizer, args, file_path=args.eval_data_file, block_size=args.block_size)
    os.makedirs(args.output_dir, exist_ok=True)
    eval_dataloader, args = get_dataloader(eval_dataset, tokenizer, args, split='eval')
    print("***** Running evaluation {} *****".format(prefix))
    print("  Num examples = {}".format(len(eval_dataset)))
    print("  Batch size = {}".format(args.eval_batch_size))
    eval_loss, nb_eval_steps = 0.0, 0
    model.eval()
    for batch in eval_dataloader:
        inputs = batch.to(args.device)
        outputs = model(inputs, labels=inputs)
        eval_loss += outputs[0].mean().item()
        nb_eval_steps += 1
    return eval_loss / nb_eval_steps


def main(argv=None):
    args = parse(SIMPLEDYG, "main_SimpleDyG.py", argv)
    args.with_mask_token = False                     # SimpleDyG tokenizer has no [MASK] (main_SimpleDyG.py:91-95)
    if args.eval_data_file is None and args.do_eval:
        raise ValueError("--eval_data_file should be specified when do_eval is true")
    if args.no_cuda or not torch.cuda.is_available():
        raise SystemExit("main_SimpleDyG: needs the MI355X (rag4dyg_amd has no CPU fallback)")
    # one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set): the evaluation's sequences are decoded
    # data-parallel (rag4dyg_amd.evaluation.get_eval_metrics); "nccl" IS RCCL on ROCm
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dev_index = int(os.environ.get("LOCAL_RANK", max(args.local_rank, 0))) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    args.device = torch.device("cuda", dev_index)
    args.n_gpu = 1
    if world > 1 and not torch.distributed.is_initialized():
        torch.distributed.init_process_group(backend=os.environ.get("R4D_DIST_BACKEND", "nccl"))
    torch.manual_seed(args.seed)
    args.para_names = ['dataset', 'method', 'time', 'nlayer', 'nhead', 'nemb', 'bz', 'lr', 'seed']
    args.para_values = [args.dataset, 'SimpleDyG', args.timestamp, args.n_layer, args.n_head, args.n_embed,
                        args.per_gpu_train_batch_size, args.learning_rate, args.seed]
    model, tokenizer, model_class, args = get_model_tokenizer(args, MODEL_CLASSES)
    if args.do_train:
        raise NotImplementedError("SimpleDyG LM training (backward pass) is outside the encode-and-retrieve hot path")
    results = {}
    if args.do_eval:
        checkpoints = [args.output_dir]
        if args.eval_all_checkpoints:
            checkpoints = list(os.path.dirname(c) for c in
                               sorted(glob.glob(args.output_dir + "/**/" + WEIGHTS_NAME, recursive=True)))
        print("Evaluate the following checkpoints: {}".format(checkpoints))
        for checkpoint in checkpoints:
            model = model_class.from_pretrained(checkpoint).to(args.device)
            loss = evaluate(args, model, tokenizer, prefix=os.path.basename(checkpoint))
            scores = get_eval_metrics(args, model, tokenizer, 0, mode="test")      # main_SimpleDyG.py:487
            results[checkpoint] = dict(eval_loss=loss, **scores)
            print(f"[{checkpoint}] eval_loss = {loss:.6f}  NDCG@5 = {scores['NDCG'][0]}  jaccard = {scores['jaccard'][0]}")
    return results


if __name__ == "__main__":
    main()
