"""Flag sets of the three stage CLIs -- names, types and defaults follow
``utils/args_parser_{SimpleDyG,retriever,generator}.py`` (the flag surface is the drop-in contract; flags
that only steer training are accepted and ignored by the encode-and-retrieve path).  Table-driven."""
import argparse

S, I, F = str, int, float
REQ = object()          # required flag
FLAG = object()         # store_true

COMMON = {
    "run_seed": FLAG, "n_gpu": (None, 1), "timestamp": (S, REQ), "dataset": (S, REQ), "train_data_file": (S, REQ),
    "output_dir": (S, REQ), "model_type": (S, REQ), "eval_data_file": (S, None), "eval_data_gt_file": (S, None),
    "test_data_file": (S, None), "test_data_gt_file": (S, None), "n_layer": (I, 12), "n_head": (I, 12),
    "n_embed": (I, 768), "node_feat_file": (S, None), "should_continue": FLAG, "model_name_or_path": (S, None),
    "config_name": (S, None), "tokenizer_name": (S, None), "cache_dir": (S, None), "block_size": (I, -1),
    "do_train": FLAG, "do_eval": FLAG, "evaluate_during_training": FLAG, "per_gpu_train_batch_size": (I, 4),
    "per_gpu_eval_batch_size": (I, 32), "gradient_accumulation_steps": (I, 1), "learning_rate": (F, 5e-5),
    "weight_decay": (F, 0.0), "adam_epsilon": (F, 1e-8), "max_grad_norm": (F, 1.0), "num_train_epochs": (F, 1.0),
    "max_steps": (I, -1), "warmup_steps": (I, 0), "logging_steps": (I, 500), "save_steps": (I, 500),
    "save_total_limit": (I, None), "eval_all_checkpoints": FLAG, "no_cuda": FLAG, "overwrite_cache": FLAG,
    "seed": (I, 42), "fp16": FLAG, "fp16_opt_level": (S, "O1"), "local_rank": (I, -1), "patience": (I, 5),
}
_FILES = {k: (S, None) for k in ("train_index_file", "train_score_file", "val_index_file", "val_score_file",
                                 "test_index_file", "test_score_file", "retrieval_checkpoint", "simpledyg_checkpoint")}
RETRIEVER = dict(COMMON, **_FILES, **{
    "train_pair_data_file": (S, None), "task": (S, "classification"), "threshold": (F, 0.5), "k": (I, 0),
    "retrieval_type": (S, "inputs"), "mlp_layers": (I, 2), "fusion": (S, "mlp"), "loss_type": (S, "cl"),
    "lambda_decay": (F, -1), "alpha": (F, 0.5), "eta": (F, 0.2), "gamma": (F, 0.5), "beta": (F, 0.2),
    "lrdecay": (I, 0), "temperature": (F, 0.07), "projector": (I, 0), "mask_file": (S, None), "freeze": FLAG,
    "weight_decay": (F, 1e-4), "warmup_steps": (I, 5), "topK": (I, 5), "m": (I, 10),
    # build extension: "full" = reference-compatible full permutation rows, "topk" = first topK indices only
    "rank_output": (S, "full"),
})
GENERATOR = dict(COMMON, **_FILES, **{
    "lrdecay": (I, 0), "mlp_layers": (I, 2), "gnn_layers": (I, 1), "fusion": (S, "mlp"), "negative": (I, 0),
    "temperature": (F, 0.1), "topK": (I, 7), "m": (I, 10), "freeze": FLAG, "weight_decay": (F, 1e-5),
})
SIMPLEDYG = dict(COMMON)


def build_parser(table, prog):
    p = argparse.ArgumentParser(prog=prog)
    for name, spec in table.items():
        if spec is FLAG:
            p.add_argument("--" + name, action="store_true")
        else:
            typ, default = spec
            kw = {}
            if typ is not None:
                kw["type"] = typ
            if default is REQ:
                kw.update(default=None, required=True)
            else:
                kw["default"] = default
            p.add_argument("--" + name, **kw)
    return p


def parse(table, prog, argv=None):
    return build_parser(table, prog).parse_args(argv)
