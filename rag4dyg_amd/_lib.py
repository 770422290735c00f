"""ctypes binding of librag4dyg_hip.so (the C ABI declared in include/r4d.h).

The product path has NO fallback: if the shared library is missing or a call fails, this raises.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int32, c_int64, c_size_t, c_uint32, c_uint64, c_void_p

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("R4D_LIB_PATH") or os.path.join(PKG, "librag4dyg_hip.so")   # override: tools/ A/B tuning only

R4D_ABI_VERSION = 6


class R4DError(RuntimeError):
    pass


class GPT2ConfigC(Structure):
    _fields_ = [("n_layer", c_int32), ("n_head", c_int32), ("n_embd", c_int32), ("vocab", c_int32),
                ("n_positions", c_int32), ("ln_eps", c_float)]


class GPT2LayerC(Structure):
    _fields_ = [(n, c_void_p) for n in ("ln_1_w", "ln_1_b", "c_attn_w", "c_attn_b", "attn_proj_w", "attn_proj_b",
                                         "ln_2_w", "ln_2_b", "c_fc_w", "c_fc_b", "mlp_proj_w", "mlp_proj_b",
                                         "c_attn_wT", "attn_proj_wT", "c_fc_wT", "mlp_proj_wT",
                                         "c_attn_w3", "attn_proj_w3", "c_fc_w3", "mlp_proj_w3",
                                         "c_attn_w3t", "attn_proj_w3t", "c_fc_w3t", "mlp_proj_w3t",
                                         "c_attn_wTg", "c_attn_lnc", "c_fc_wTg", "c_fc_lnc",
                                         "c_attn_h2", "attn_proj_h2", "c_fc_h2", "mlp_proj_h2")]


class GreedyStateC(Structure):
    _fields_ = [(n, c_void_p) for n in ("last_d", "logits_d", "next_d", "lens_d", "pos_d", "active_d", "gen_len_d",
                                         "out_tokens_d", "params_d")] + [("out_cap", c_int32)]


class GPT2LayerGradsC(Structure):
    _fields_ = [(n, c_void_p) for n in ("ln_1_w", "ln_1_b", "c_attn_w", "c_attn_b", "attn_proj_w", "attn_proj_b",
                                         "ln_2_w", "ln_2_b", "c_fc_w", "c_fc_b", "mlp_proj_w", "mlp_proj_b")]


class TrainDropoutC(Structure):                 # r4d_train_dropout
    _fields_ = [("embd_p", c_float), ("attn_p", c_float), ("resid_p", c_float), ("seed", c_uint64), ("step", c_uint64)]


class GPT2GradsC(Structure):
    _fields_ = [("wte", c_void_p), ("wpe", c_void_p), ("ln_f_w", c_void_p), ("ln_f_b", c_void_p),
                ("layers", POINTER(GPT2LayerGradsC))]


class GPT2WeightsC(Structure):
    _fields_ = [("wte", c_void_p), ("wpe", c_void_p), ("ln_f_w", c_void_p), ("ln_f_b", c_void_p),
                ("layers", POINTER(GPT2LayerC)), ("lm_head", c_void_p)]


_P = c_void_p
# name -> (restype, argtypes); one entry per symbol declared in include/r4d.h
PROTOTYPES = {
    "r4d_abi_version": (c_int32, []),
    "r4d_build_flags": (c_int32, []),
    "r4d_last_error": (c_char_p, []),
    "r4d_gpt2_workspace_bytes": (c_size_t, [POINTER(GPT2ConfigC), c_int32, c_int32]),
    "r4d_gpt2_encode_f32": (c_int32, [POINTER(GPT2ConfigC), POINTER(GPT2WeightsC), _P, _P, c_int32, c_int32,
                                      _P, _P, _P, _P, _P, c_size_t, _P]),
    "r4d_gpt2_groups_workspace_bytes": (c_size_t, [POINTER(GPT2ConfigC), c_int32, POINTER(c_int32), POINTER(c_int32)]),
    "r4d_gpt2_encode_groups_f32": (c_int32, [POINTER(GPT2ConfigC), POINTER(GPT2WeightsC), c_int32, POINTER(_P),
                                             POINTER(c_int32), POINTER(c_int32), _P, _P, c_size_t, _P]),
    "r4d_gpt2_encode_groups_ex_f32": (c_int32, [POINTER(GPT2ConfigC), POINTER(GPT2WeightsC), c_int32, POINTER(_P), POINTER(_P),
                                                POINTER(c_int32), POINTER(c_int32), _P, _P, _P, _P, c_size_t, _P]),
    "r4d_set_attention_fused": (c_int32, [c_int32]),
    "r4d_set_attention_h2": (c_int32, [c_int32]),
    "r4d_set_attention_kblk": (c_int32, [c_int32]),
    "r4d_pack_kblk_words": (c_int32, [_P, c_int64, c_int32, c_int32, _P, _P]),
    "r4d_attention_h2_kblk_f32": (c_int32, [_P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "r4d_set_range_flag": (c_int32, [_P]),
    "r4d_set_gemm_h2p": (c_int32, [c_int32]),
    "r4d_layernorm_lines_f32": (c_int32, [_P, _P, _P, c_int32, c_int32, c_float, _P, _P]),
    "r4d_split2_lines_f16": (c_int32, [_P, c_int64, c_int32, _P, _P]),
    "r4d_conv1d_h2p_f32": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "r4d_pack_h2_words_f32": (c_int32, [_P, c_int64, _P, _P]),
    "r4d_attention_h2_f32": (c_int32, [_P, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "r4d_gpt2_decode_workspace_bytes": (c_size_t, [POINTER(GPT2ConfigC), c_int32]),
    "r4d_gpt2_decode_step_f32": (c_int32, [POINTER(GPT2ConfigC), POINTER(GPT2WeightsC), _P, _P, _P, _P, c_int32, c_int32,
                                           _P, _P, c_size_t, _P]),
    "r4d_gpt2_greedy_workspace_bytes": (c_size_t, [POINTER(GPT2ConfigC), c_int32]),
    "r4d_gpt2_greedy_step_f32": (c_int32, [POINTER(GPT2ConfigC), POINTER(GPT2WeightsC), POINTER(GreedyStateC), _P, c_int32,
                                           c_int32, _P, c_size_t, _P]),
    "r4d_gpt2_greedy_graph_create": (c_int32, [POINTER(GPT2ConfigC), POINTER(GPT2WeightsC), POINTER(GreedyStateC), _P,
                                               c_int32, c_int32, _P, c_size_t, POINTER(_P)]),
    "r4d_decode_graph_launch": (c_int32, [_P, c_int32, _P]),
    "r4d_decode_graph_destroy": (None, [_P]),
    "r4d_lm_logits_f32": (c_int32, [_P, _P, c_int32, c_int32, c_int32, _P, _P]),
    "r4d_layernorm_f32": (c_int32, [_P, _P, _P, c_int32, c_int32, c_float, _P, _P]),
    "r4d_conv1d_f32": (c_int32, [_P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "r4d_split3_planes_bf16": (c_int32, [_P, c_int32, c_int32, c_int32, _P, _P]),
    "r4d_conv1d_s3_f32": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "r4d_split2_planes_f16": (c_int32, [_P, c_int32, c_int32, c_int32, _P, _P]),
    "r4d_conv1d_h2_f32": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "r4d_fold_layernorm_f32": (c_int32, [_P, _P, _P, c_int32, c_int32, _P, _P, _P]),
    "r4d_set_gemm_split3": (c_int32, [c_int32]),
    "r4d_get_gemm_split3": (c_int32, []),
    "r4d_attention_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "r4d_attention_f32": (c_int32, [_P, c_int32, c_int32, c_int32, c_int32, _P, _P, c_size_t, _P]),
    "r4d_normalize_rows_f32": (c_int32, [_P, c_int32, c_int32, _P, _P]),
    "r4d_score_topk_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "r4d_score_topk_f32": (c_int32, [_P, _P, c_int32, c_int32, c_int32, c_int32, c_int64, _P, _P, _P, _P,
                                     c_size_t, _P]),
    "r4d_topk_f32_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "r4d_topk_f32": (c_int32, [_P, c_int32, c_int32, c_int32, _P, _P, _P, c_size_t, _P]),
    "r4d_merge_topk_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "r4d_merge_topk_f32": (c_int32, [_P, _P, c_int32, c_int32, c_int32, _P, _P, _P, c_size_t, _P]),
    "r4d_argsort_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "r4d_argsort_desc_f32": (c_int32, [_P, c_int32, c_int32, _P, _P, c_size_t, _P]),
    "r4d_argsort_desc_f64": (c_int32, [_P, c_int32, c_int32, _P, _P, c_size_t, _P]),
    "r4d_jaccard_f64": (c_int32, [_P, _P, c_int32, c_int32, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "r4d_jaccard_ordered_f64": (c_int32, [_P, _P, c_int32, c_int32, _P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P]),
    "r4d_jaccard_prepared_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "r4d_jaccard_prepared_f64": (c_int32, [_P, _P, c_int32, c_int32, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P,
                                           c_size_t, _P]),
    "r4d_topk_f64_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "r4d_topk_f64": (c_int32, [_P, c_int32, c_int32, c_int32, _P, _P, _P, c_size_t, _P]),
    "r4d_gpt2_train_workspace_bytes": (c_size_t, [POINTER(GPT2ConfigC), c_int32, POINTER(c_int32), POINTER(c_int32)]),
    "r4d_gpt2_train_forward_f32": (c_int32, [POINTER(GPT2ConfigC), POINTER(GPT2WeightsC), c_int32, POINTER(_P), POINTER(c_int32),
                                             POINTER(c_int32), _P, POINTER(TrainDropoutC), _P, c_size_t, _P]),
    "r4d_gpt2_train_backward_f32": (c_int32, [POINTER(GPT2ConfigC), POINTER(GPT2WeightsC), POINTER(GPT2GradsC), c_int32, POINTER(_P),
                                              POINTER(c_int32), POINTER(c_int32), _P, POINTER(TrainDropoutC), _P, c_size_t, _P]),
    "r4d_retriever_losses_workspace_bytes": (c_size_t, [c_int32]),
    "r4d_retriever_losses_f32": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_float, c_float, c_float, c_float, _P, _P, _P, c_size_t, _P]),
    "r4d_dropout_f32": (c_int32, [_P, _P, c_int64, _P, c_float, c_uint64, c_uint64, c_uint32, c_uint64, _P]),
    "r4d_weight_grad_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "r4d_weight_grad_f32": (c_int32, [_P, _P, c_int32, c_int32, c_int32, _P, _P, _P, c_size_t, _P]),
    "r4d_layernorm_bwd_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "r4d_layernorm_bwd_f32": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_float, _P, _P, _P, _P, c_size_t, _P]),
    "r4d_gelu_new_f32": (c_int32, [_P, c_int64, _P, _P]),
    "r4d_gelu_new_bwd_f32": (c_int32, [_P, _P, c_int64, _P, _P]),
    "r4d_causal_softmax_bwd_f32": (c_int32, [_P, _P, c_int32, c_int32, c_int32, c_float, _P]),
    "r4d_sumsq_accumulate_f32": (c_int32, [_P, c_int64, _P, _P]),
    "r4d_adamw_step_f32": (c_int32, [_P, _P, _P, _P, c_int64, c_double, c_double, c_double, c_double, c_double, c_int32, _P, c_float, _P]),
    "r4d_dispatch_num_branches": (c_int32, []),
    "r4d_dispatch_branch_name": (c_char_p, [c_int32]),
    "r4d_dispatch_branch_hits": (c_int64, [c_int32]),
    "r4d_dispatch_reset": (c_int32, []),
    "r4d_profile_enable": (c_int32, [c_int32]),
    "r4d_profile_num_classes": (c_int32, []),
    "r4d_profile_class_name": (c_char_p, [c_int32]),
    "r4d_profile_read": (c_int32, [c_int32, POINTER(c_double), POINTER(c_int64), POINTER(c_double)]),
}

_LIB = None


def load():
    """Load (once) and return the ctypes handle; raises R4DError when the HIP extension is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise R4DError(f"{LIB_PATH} not found: build it with `python -m rag4dyg_amd.build` "
                       "(there is no CPU fallback for the product path)")
    import torch  # noqa: F401  -- torch's bundled libamdhip64 must be the HIP runtime both sides share
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    got = lib.r4d_abi_version()
    if got != R4D_ABI_VERSION:
        raise R4DError(f"ABI mismatch: library {got}, binding {R4D_ABI_VERSION}")
    flags = lib.r4d_build_flags()
    if flags and os.environ.get("R4D_ALLOW_ABLATED_LIB") != "1":      # tools/kc_ablate.sh builds: wrong results by construction
        raise R4DError(f"{LIB_PATH} is a kernel-ablation build (r4d_build_flags() = {flags:#x}): timing aid for tools/ only; "
                       "the product refuses it (set R4D_ALLOW_ABLATED_LIB=1 in a tuning script)")
    _LIB = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().r4d_last_error().decode("utf-8", "replace")
        raise R4DError(f"{what} failed (rc={rc}): {msg}")
