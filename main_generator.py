#!/usr/bin/env python3
"""Drop-in for the reference ``main_generator.py`` (flags of ``utils/args_parser_generator.py``), INFERENCE side:
``--do_eval`` loads every generator checkpoint under ``--output_dir`` (GPT-2 + ``gnn_fusion`` / ``mlp_fusion`` weights,
``main_generator.py:108-127``) and runs the reference's RAG evaluation -- graph-pooling or MLP fusion of the top-K
retrieved training sequences (``*_index.gen`` written by this build's ``main_retriever.py``), greedy link prediction,
R@5 / NDCG@5 / Jaccard (``utils/Evaluation_generator.py:49-265``) -- on the MI355X kernels; launched with
``python -m torch.distributed.run --nproc-per-node N`` the test queries are decoded data-parallel, one process per GPU.
Generator TRAINING (``train/train_generator.py``, backward pass) is not part of this build and raises."""
import glob
import os

import torch
import torch.distributed

from rag4dyg_amd.cli_args import GENERATOR, parse
from rag4dyg_amd.generator import get_eval_metrics_generator
from rag4dyg_amd.gpt2 import GPT2Config, GPT2LMHeadModelRAG
from rag4dyg_amd.tokenizer import WordLevelTokenizer, get_model_tokenizer

WEIGHTS_NAME = "pytorch_model.bin"
MODEL_CLASSES = {"gpt2": (GPT2Config, GPT2LMHeadModelRAG, WordLevelTokenizer)}


def run_name_of(args):
    """``main_generator.py:54-66`` (``retrieval_type`` is read there but not defined by the generator's parser)."""
    args.para_names = ['d', 'nl', 'nh', 'nd', 'bz', 'lr', 'se', 'fus', 'm', 'k', 'mlp', 'gnn', 'lrdecay', 'wd', 'wm', 're']
    args.para_values = [args.dataset, args.n_layer, args.n_head, args.n_embed, args.per_gpu_train_batch_size,
                        args.learning_rate, args.seed, args.fusion, args.m, args.topK, args.mlp_layers, args.gnn_layers,
                        args.lrdecay, args.weight_decay, args.warmup_steps, getattr(args, "retrieval_type", "")]
    return ''.join(f"{n}:{v}_" for n, v in zip(args.para_names, args.para_values))


def main(argv=None):
    args = parse(GENERATOR, "main_generator.py", argv)
    args.with_mask_token = False                     # utils/tokenizer_generator.py:46-88 adds no [MASK]
    if args.eval_data_file is None and args.do_eval:
        raise ValueError("--eval_data_file should be specified when do_eval is true")
    if args.no_cuda or not torch.cuda.is_available():
        raise SystemExit("main_generator: needs the MI355X (rag4dyg_amd has no CPU fallback)")
    # one process per GPU: under `python -m torch.distributed.run --nproc-per-node N main_generator.py ...` the test
    # queries are decoded data-parallel (generator.get_eval_metrics_generator); "nccl" IS RCCL on ROCm
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", max(args.local_rank, 0)))
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    args.device = torch.device("cuda", dev_index)
    args.n_gpu = 1
    if world > 1 and not torch.distributed.is_initialized():
        torch.distributed.init_process_group(backend=os.environ.get("R4D_DIST_BACKEND", "nccl"))
    rank0 = not torch.distributed.is_initialized() or torch.distributed.get_rank() == 0
    torch.manual_seed(args.seed)
    args.run_name = run_name_of(args)
    model, tokenizer, model_class, args = get_model_tokenizer(args, MODEL_CLASSES)
    if args.fusion == "mlp":
        model.get_mlp(512, args.m, args.mlp_layers)                                   # main_generator.py:81-82
    if "graphpooling" in args.fusion:
        model.get_gnn(args.n_embed, int(args.n_embed / 2), args.n_embed, args.gnn_layers, 0.2)      # :84-85
    if args.do_train:
        raise NotImplementedError("generator training (GNN/MLP fusion + LM head, backward pass) is outside this build; "
                                  "train with the reference and evaluate the checkpoints here")
    results = {}
    if args.do_eval:
        checkpoints = [args.output_dir]
        if args.eval_all_checkpoints:
            checkpoints = list(os.path.dirname(c) for c in
                               sorted(glob.glob(args.output_dir + "/**/" + WEIGHTS_NAME, recursive=True)))
        if rank0:
            print("Evaluate the following checkpoints: {}".format(checkpoints))
        for checkpoint in checkpoints:
            global_step = checkpoint.split("-")[-1] if len(checkpoints) > 1 else ""
            state_dict = torch.load(os.path.join(checkpoint, WEIGHTS_NAME), map_location="cpu", weights_only=True)
            model.load_state_dict(state_dict)          # :118-119 (strict, like upstream); an untied checkpoint stays untied
            model.to(args.device)
            results[checkpoint] = get_eval_metrics_generator(args, 0, model, tokenizer, global_step, mode="test",
                                                             is_rag=True)
            if rank0:
                print(f"[{checkpoint}] {results[checkpoint]}")
    return results


if __name__ == "__main__":
    main()
