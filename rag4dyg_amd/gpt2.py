"""Host-side mirror of the reference's GPT-2 call surface, running on the gfx950 kernels.

Same class names, argument meaning, return structure, error behaviour and state-dict / config.json
layout as the reference (``models/modeling_gpt2.py``, ``models/modeling_rag.py``,
``models/modeling_utils.py``, ``models/configuration_gpt2.py``) so that reference checkpoints load
unchanged and the reference call sites (``train/train_retriever.py:419,430``,
``main_SimpleDyG.py:168``, ``utils/model.py:161,222``) keep working.  The modules only HOLD parameters;
``forward`` hands their device pointers to ``r4d_gpt2_encode_f32`` -- inference only (no autograd), GPU
only (CPU tensors raise: there is no fallback).
"""
import contextlib
import ctypes
import weakref
import json
import os

import torch
import torch.nn as nn

from . import _lib, ops


# The cached decode step reads LayerNorm-folded copies of c_attn / c_fc (ops.fold_layernorm, +16 MB per layer at d = 768, made once per
# checkpoint).  False (or R4D_DECODE_FOLD_LN=0): the step forms the LayerNorm constants per launch from the plain k-contiguous copies.
FOLD_DECODE_LAYERNORM = os.environ.get("R4D_DECODE_FOLD_LN", "1") != "0"

class GPT2Config:
    """``models/configuration_gpt2.py:120-162`` defaults + the keys the reference adds (``utils/tokenizer.py:21-26,51``)."""

    model_type = "gpt2"

    def __init__(self, vocab_size=50257, n_positions=1024, n_ctx=1024, n_embd=768, n_layer=12, n_head=12,
                 resid_pdrop=0.1, embd_pdrop=0.1, attn_pdrop=0.1, layer_norm_epsilon=1e-5, initializer_range=0.02,
                 **kwargs):
        self.vocab_size = vocab_size
        self.n_positions = n_positions
        self.n_ctx = n_ctx
        self.n_embd = n_embd
        self.n_layer = n_layer
        self.n_head = n_head
        self.resid_pdrop = resid_pdrop
        self.embd_pdrop = embd_pdrop
        self.attn_pdrop = attn_pdrop
        self.layer_norm_epsilon = layer_norm_epsilon
        self.initializer_range = initializer_range
        self.output_past = kwargs.pop("output_past", True)           # configuration_utils.py:60-65
        self.output_hidden_states = kwargs.pop("output_hidden_states", False)
        self.output_attentions = kwargs.pop("output_attentions", False)
        for k, v in kwargs.items():
            setattr(self, k, v)

    def to_dict(self):
        return dict(self.__dict__, model_type=self.model_type)

    def save_pretrained(self, save_directory):
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2, sort_keys=True)

    @classmethod
    def from_pretrained(cls, path, **kwargs):
        cfg_file = os.path.join(path, "config.json") if os.path.isdir(path) else path
        if not os.path.isfile(cfg_file):
            raise OSError(f"config file not found: {cfg_file} (no network: only local paths are supported)")
        with open(cfg_file) as f:
            d = json.load(f)
        d.pop("model_type", None)
        d.update(kwargs)
        return cls(**d)


_SKIP_INIT = [False]


@contextlib.contextmanager
def skip_random_init():
    """Construct a model WITHOUT drawing its random initial weights -- for callers that overwrite every parameter right afterwards
    (the evaluation-only command lines: strict ``load_state_dict`` of a checkpoint).  The draws were 0.46 s of a 0.56 s
    ``main_retriever.py --do_eval`` run (each tensor initialised twice on the CPU, by its torch module and by ``_init_weights``,
    then replaced by the checkpoint).  Every weight that WOULD have been drawn is filled with NaN instead (a memset, not a
    draw): a tensor a later non-strict or partial load leaves out turns every output NaN, which the range guard reports
    (``ops.check_range``) -- it cannot evaluate quietly on arbitrary bytes.  Not thread-safe: ``torch.nn.init`` is
    patched process-wide for the duration of the ``with`` block; construct models on one thread."""
    saved = (nn.init.normal_, nn.init.kaiming_uniform_, nn.init.uniform_)

    def _nan_fill(t, *a, **k):
        with torch.no_grad():
            return t.fill_(float("nan"))
    nn.init.normal_ = nn.init.kaiming_uniform_ = nn.init.uniform_ = _nan_fill
    _SKIP_INIT[0] = True
    try:
        yield
    finally:
        _SKIP_INIT[0] = False
        nn.init.normal_, nn.init.kaiming_uniform_, nn.init.uniform_ = saved


class Conv1D(nn.Module):
    """Parameter holder with the reference layout: weight [nx, nf] (``modeling_utils.py:1255-1265``)."""

    def __init__(self, nf, nx):
        super().__init__()
        self.nf = nf
        self.weight = nn.Parameter(torch.full((nx, nf), float("nan")) if _SKIP_INIT[0] else torch.empty(nx, nf).normal_(std=0.02))
        self.bias = nn.Parameter(torch.zeros(nf))

    def forward(self, x):
        return ops.conv1d(x.contiguous(), self.weight, self.bias)


class Attention(nn.Module):
    def __init__(self, nx, n_ctx, config):
        super().__init__()
        assert nx % config.n_head == 0                                # modeling_gpt2.py:105
        # causal buffer kept so that state_dict() has the reference's keys (modeling_gpt2.py:106); unused by the kernels
        self.register_buffer("bias", torch.tril(torch.ones(n_ctx, n_ctx)).view(1, 1, n_ctx, n_ctx))
        self.n_head = config.n_head
        self.c_attn = Conv1D(nx * 3, nx)
        self.c_proj = Conv1D(nx, nx)


class MLP(nn.Module):
    def __init__(self, n_state, config):
        super().__init__()
        self.c_fc = Conv1D(n_state, config.n_embd)
        self.c_proj = Conv1D(config.n_embd, n_state)


class Block(nn.Module):
    def __init__(self, n_ctx, config):
        super().__init__()
        nx = config.n_embd
        self.ln_1 = nn.LayerNorm(nx, eps=config.layer_norm_epsilon)
        self.attn = Attention(nx, n_ctx, config)
        self.ln_2 = nn.LayerNorm(nx, eps=config.layer_norm_epsilon)
        self.mlp = MLP(4 * nx, config)


class _PreTrained(nn.Module):
    config_class = GPT2Config
    base_model_prefix = "transformer"

    def _init_weights(self, module):
        """``modeling_gpt2.py:251-262``."""
        if isinstance(module, (nn.Linear, nn.Embedding, Conv1D)):
            if not _SKIP_INIT[0]:
                module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
            else:
                module.weight.data.fill_(float("nan"))                 # loud if no checkpoint tensor replaces it
            if isinstance(module, (nn.Linear, Conv1D)) and module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)

    def save_pretrained(self, save_directory):
        """``modeling_utils.py:277-297``: config.json + pytorch_model.bin (= torch.save(state_dict))."""
        assert os.path.isdir(save_directory), "Saving path should be a directory where the model and configuration can be saved"
        self.config.architectures = [self.__class__.__name__]
        self.config.save_pretrained(save_directory)
        torch.save({k: v.cpu() for k, v in self.state_dict().items()}, os.path.join(save_directory, "pytorch_model.bin"))

    @classmethod
    def from_pretrained(cls, path, config=None, **kwargs):
        """Local-directory subset of ``modeling_utils.py:300-582``: loads ``pytorch_model.bin`` and maps the
        ``transformer.`` prefix either way (base model <- LM-head checkpoint and vice versa, :530-541)."""
        if config is None:
            config = cls.config_class.from_pretrained(path)
        model = cls(config)
        f = os.path.join(path, "pytorch_model.bin")
        if not os.path.isfile(f):
            raise OSError(f"Error no file named pytorch_model.bin found in directory {path}")
        sd = torch.load(f, map_location="cpu", weights_only=True)
        own = model.state_dict()
        has_prefix_ckpt = any(k.startswith("transformer.") for k in sd)
        has_prefix_own = any(k.startswith("transformer.") for k in own)
        if has_prefix_ckpt and not has_prefix_own:
            sd = {k[len("transformer."):]: v for k, v in sd.items() if k.startswith("transformer.")}
        elif has_prefix_own and not has_prefix_ckpt:
            sd = {"transformer." + k: v for k, v in sd.items()}
        # modeling_utils.py:543-566: size mismatches and missing weights are errors, not silent re-initialisation.  Missing
        # keys are tolerated only where the reference tolerates them: the causal `attn.bias` buffers, `lm_head.weight`
        # (tied to wte right below) and the fusion modules a generator checkpoint may or may not carry.
        bad = [f"size mismatch for {k}: checkpoint {tuple(v.shape)} vs model {tuple(own[k].shape)}"
               for k, v in sd.items() if k in own and own[k].shape != v.shape]
        missing = [k for k in own if k not in sd and not k.endswith(".attn.bias") and k != "lm_head.weight"
                   and not k.startswith(("mlp_fusion.", "gnn_fusion."))]
        if bad or missing:
            raise RuntimeError("Error(s) in loading state_dict for {}:\n\t{}".format(
                cls.__name__, "\n\t".join(bad + [f"missing key {k}" for k in missing])))
        sd = {k: v for k, v in sd.items() if k in own}
        model.load_state_dict(sd, strict=False)
        if hasattr(model, "tie_weights") and not getattr(model, "_untied_by_checkpoint", False):
            model.tie_weights()                                   # modeling_utils.py:571
        model.eval()
        return model


# Parameters written through RAW POINTERS (``r4d_adamw_step_f32``: training.AdamW.step) do not bump torch's in-place version
# counter, which is what the derived-weight caches below are keyed on.  Every such writer calls ``note_raw_parameter_write()``;
# the generation is part of every cache stamp, so transposed copies, bf16x3 planes and LayerNorm-folded decode weights of ALL
# models are rebuilt on their next use (evaluation between optimizer steps used stale planes).
_RAW_WRITE_GENERATION = [0]


def note_raw_parameter_write():
    _RAW_WRITE_GENERATION[0] += 1


class GPT2Model(_PreTrained):
    """``models/modeling_gpt2.py:328-509``."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.output_hidden_states = config.output_hidden_states
        self.output_past = config.output_past
        self.wte = nn.Embedding(config.vocab_size, config.n_embd)
        self.wpe = nn.Embedding(config.n_positions, config.n_embd)
        self.h = nn.ModuleList([Block(config.n_ctx, config) for _ in range(config.n_layer)])
        self.ln_f = nn.LayerNorm(config.n_embd, eps=config.layer_norm_epsilon)
        self.apply(self._init_weights)

    def get_input_embeddings(self):
        return self.wte

    def set_input_embeddings(self, new_embeddings):
        self.wte = new_embeddings

    def resize_token_embeddings(self, new_num_tokens=None):
        """``modeling_utils.py:183-248``: grow/shrink wte keeping the first min(old,new) rows."""
        old = self.wte
        if new_num_tokens is None or new_num_tokens == old.num_embeddings:
            return old
        new = nn.Embedding(new_num_tokens, old.embedding_dim).to(old.weight.device)
        if not _SKIP_INIT[0]:
            new.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        else:
            new.weight.data.fill_(float("nan"))
        n = min(old.num_embeddings, new_num_tokens)
        new.weight.data[:n, :] = old.weight.data[:n, :]
        self.wte = new
        self.config.vocab_size = new_num_tokens
        return new

    # ------------------------------------------------------------------ kernel hand-off
    # per-object caches and the back-link to an owning LM-head model: never copied or pickled (copy.deepcopy(model) -- the
    # reference's `best_model = copy.deepcopy(model)` -- and torch.save(model) go through __getstate__); an owner re-links
    # its own copy in _LMHeadBase.__setstate__
    _TRANSIENT = ("_wt_cache", "_w3_cache", "_h2_cache", "_fold_cache", "_greedy_decoders", "_lm_head_weight")

    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k not in self._TRANSIENT}

    def _wt(self, w):
        """Contiguous transposed copy [out,in] of a static Conv1D weight, cached until the weight changes
        (keyed by storage pointer and in-place version counter)."""
        cache = self.__dict__.setdefault("_wt_cache", {})
        key = id(w)
        ent = cache.get(key)
        stamp = (w.data_ptr(), w._version, _RAW_WRITE_GENERATION[0])
        if ent is None or ent[0] != stamp:
            ent = (stamp, w.detach().t().contiguous())
            cache[key] = ent
        return ent[1].data_ptr()

    def _w3(self, w):
        """bf16x3 planes [3,out,in] of a static Conv1D weight (``ops.split3_planes``), cached like ``_wt``; None when the
        split GEMM is switched off (``ops.set_gemm_split3(False)`` / ``R4D_GEMM_SPLIT3=0``) or the shape has no kernel."""
        if not ops.gemm_split3_enabled() or w.shape[0] % 32 != 0:
            return None
        cache = self.__dict__.setdefault("_w3_cache", {})
        key = id(w)
        ent = cache.get(key)
        stamp = (w.data_ptr(), w._version, _RAW_WRITE_GENERATION[0])
        if ent is None or ent[0] != stamp:
            ent = (stamp, ops.split3_planes(w.detach()))
            cache[key] = ent
        return ent[1].data_ptr()

    def _h2(self, w):
        """f16x2 lines [out, in/32, 2, 32] of a static Conv1D weight (``ops.split2_planes``), cached like ``_wt``; None unless
        ``ops.gemm_mode() == "f16x2"``, and for a weight outside the fp16 range or a shape without a kernel."""
        if ops.gemm_mode() != "f16x2" or w.shape[0] % 32 != 0:
            return None
        cache = self.__dict__.setdefault("_h2_cache", {})
        key = id(w)
        ent = cache.get(key)
        stamp = (w.data_ptr(), w._version, _RAW_WRITE_GENERATION[0])
        if ent is None or ent[0] != stamp:
            ent = (stamp, ops.split2_planes(w.detach()))
            cache[key] = ent
        return ent[1].data_ptr() if ent[1] is not None else None

    def _fold(self, w, ln):
        """Decode-only (pointer to gain-folded [out,in] copy, pointer to its [2,out] column constants) of a Conv1D weight that
        reads LayerNorm ``ln`` (``ops.fold_layernorm``), cached until the weight or the LayerNorm changes."""
        cache = self.__dict__.setdefault("_fold_cache", {})
        key = id(w)
        stamp = (w.data_ptr(), w._version, ln.weight.data_ptr(), ln.weight._version, ln.bias.data_ptr(), ln.bias._version,
                 _RAW_WRITE_GENERATION[0])
        ent = cache.get(key)
        if ent is None or ent[0] != stamp:
            self._wt(w)
            wT = self._wt_cache[id(w)][1]
            ent = (stamp, ops.fold_layernorm(wT, ln.weight.detach(), ln.bias.detach()))
            cache[key] = ent
        return ent[1][0].data_ptr(), ent[1][1].data_ptr()

    def _c_structs(self, decode=False):
        cfg = self.config
        c = _lib.GPT2ConfigC(cfg.n_layer, cfg.n_head, cfg.n_embd, self.wte.num_embeddings, self.wpe.num_embeddings,
                             cfg.layer_norm_epsilon)
        layers = (_lib.GPT2LayerC * cfg.n_layer)()

        def p(t):
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
                raise _lib.R4DError("model weights must be contiguous fp32 tensors on the GPU "
                                    "(call model.to('cuda'); there is no CPU fallback)")
            return t.data_ptr()
        for i, blk in enumerate(self.h):
            ws4 = (blk.attn.c_attn.weight, blk.attn.c_proj.weight, blk.mlp.c_fc.weight, blk.mlp.c_proj.weight)
            h2 = [self._h2(w) for w in ws4]                                       # f16x2 mode only (None otherwise / out of fp16 range)
            w3 = [self._w3(w) if h is None else None for w, h in zip(ws4, h2)]    # bf16x3 planes: that mode, or the f16x2 fallback
            layers[i] = _lib.GPT2LayerC(p(blk.ln_1.weight), p(blk.ln_1.bias), p(blk.attn.c_attn.weight),
                                        p(blk.attn.c_attn.bias), p(blk.attn.c_proj.weight), p(blk.attn.c_proj.bias),
                                        p(blk.ln_2.weight), p(blk.ln_2.bias), p(blk.mlp.c_fc.weight),
                                        p(blk.mlp.c_fc.bias), p(blk.mlp.c_proj.weight), p(blk.mlp.c_proj.bias),
                                        self._wt(blk.attn.c_attn.weight), self._wt(blk.attn.c_proj.weight),
                                        self._wt(blk.mlp.c_fc.weight), self._wt(blk.mlp.c_proj.weight), *w3)
            layers[i].c_attn_h2, layers[i].attn_proj_h2, layers[i].c_fc_h2, layers[i].mlp_proj_h2 = h2
            if decode and FOLD_DECODE_LAYERNORM and cfg.n_embd in (512, 768):   # the cached decode step's LayerNorm-fused projections (ABI v4)
                layers[i].c_attn_wTg, layers[i].c_attn_lnc = self._fold(blk.attn.c_attn.weight, blk.ln_1)
                layers[i].c_fc_wTg, layers[i].c_fc_lnc = self._fold(blk.mlp.c_fc.weight, blk.ln_2)
        head = self.__dict__.get("_lm_head_weight")              # set by an LM-head model whose lm_head is NOT tied to wte
        head = head() if head is not None else None
        w = _lib.GPT2WeightsC(p(self.wte.weight), p(self.wpe.weight), p(self.ln_f.weight), p(self.ln_f.bias), layers,
                              p(head) if head is not None and head.data_ptr() != self.wte.weight.data_ptr() else None)
        return c, w, layers

    def _range_guarded(self, device, what, thunk):
        """Run ``thunk()`` (one or more ``encode*`` calls) under the range guard (``include/r4d.h``, ABI v6): ONE read of the device
        word afterwards; a non-finite hidden state in gemm mode "f16x2" (an activation beyond fp16's exponent range) re-runs the thunk
        once under "bf16x3" with a RuntimeWarning, and ``R4DError`` if it comes up again or in any other mode."""
        ops.range_flag(device).zero_()
        out = thunk()
        flag = ops.take_range_flag()
        if flag & ops.RANGE_NONFINITE_HIDDEN and ops.gemm_mode() == "f16x2":
            import warnings
            warnings.warn(f"rag4dyg_amd: an activation left the fp16 range of the f16x2 arithmetic (non-finite hidden state); "
                          f"re-running {what} with the bf16x3 GEMMs", RuntimeWarning, stacklevel=3)
            ops.set_gemm_mode("bf16x3")
            try:
                out = thunk()
                flag = ops.take_range_flag()
            finally:
                ops.set_gemm_mode("f16x2")
        if flag & ops.RANGE_NONFINITE_HIDDEN:
            raise _lib.R4DError(f"{what}: a non-finite hidden state reached ln_f (weights or inputs overflow fp32 itself, or hold NaN)")
        return out

    @torch.no_grad()
    def encode(self, input_ids=None, inputs_embeds=None, want_hidden=True, want_meanpool=False, want_layers=False,
               want_qkv=False):
        """One fused encoder pass.  Returns dict(hidden [B,T,d], meanpool [B,d], layers [L,B,T,d], qkv [L,B,T,3d])."""
        if input_ids is not None and inputs_embeds is not None:
            raise ValueError("You cannot specify both input_ids and inputs_embeds at the same time")
        if input_ids is None and inputs_embeds is None:
            raise ValueError("You have to specify either input_ids or inputs_embeds")
        src = input_ids if input_ids is not None else inputs_embeds
        if not src.is_cuda:
            raise _lib.R4DError("rag4dyg_amd runs on the GPU only: move inputs to 'cuda' (no CPU fallback)")
        dev = src.device
        d = self.config.n_embd
        if input_ids is not None:
            input_ids = input_ids.view(-1, input_ids.shape[-1]).to(torch.int64).contiguous()
            B, T = input_ids.shape
        else:
            inputs_embeds = inputs_embeds.to(torch.float32).contiguous()
            B, T = inputs_embeds.shape[:2]
        L = self.config.n_layer
        lib = _lib.load()
        ops.range_flag(dev)                                           # range guard registered (include/r4d.h, ABI v6)
        c, w, _keep = self._c_structs()
        ws = ops.workspace(lib.r4d_gpt2_workspace_bytes(ctypes.byref(c), B, T), dev, "gpt2")
        out = {}
        hidden = torch.empty(B, T, d, dtype=torch.float32, device=dev) if want_hidden else None
        pool = torch.empty(B, d, dtype=torch.float32, device=dev) if want_meanpool else None
        layers = torch.empty(L, B, T, d, dtype=torch.float32, device=dev) if want_layers else None
        qkv = torch.empty(L, B, T, 3 * d, dtype=torch.float32, device=dev) if want_qkv else None

        def ptr(t):
            return t.data_ptr() if t is not None else None
        _lib.check(lib.r4d_gpt2_encode_f32(ctypes.byref(c), ctypes.byref(w), ptr(input_ids), ptr(inputs_embeds), B, T,
                                           ptr(hidden), ptr(pool), ptr(layers), ptr(qkv), ws.data_ptr(), ws.numel(),
                                           torch.cuda.current_stream().cuda_stream), "gpt2_encode")
        out.update(hidden=hidden, meanpool=pool, layers=layers, qkv=qkv)
        return out

    # ------------------------------------------------------------------ incremental decode (key/value cache)
    def new_kv_cache(self, B, t_cap, device):
        """[n_layer, B, t_cap, 2d] cache for ``decode_step`` (K row then V row per position)."""
        return torch.empty(self.config.n_layer, B, t_cap, 2 * self.config.n_embd, dtype=torch.float32, device=device)

    @torch.no_grad()
    def prefill(self, kv_cache, input_ids=None, inputs_embeds=None):
        """Full forward over a right-padded batch [B,T] that also fills rows [0,T) of ``kv_cache`` (the K / V columns of
        every layer's c_attn output -- the model's ``presents``, ``modeling_gpt2.py:187``).  Returns hidden [B,T,d]."""
        src = input_ids if input_ids is not None else inputs_embeds
        r = self._range_guarded(src.device, "prefill", lambda: self.encode(input_ids, inputs_embeds, want_hidden=True, want_qkv=True))
        d = self.config.n_embd
        T = r["qkv"].shape[2]
        if T > kv_cache.shape[2] or r["qkv"].shape[1] != kv_cache.shape[1]:
            raise ValueError(f"prefill: batch {tuple(r['qkv'].shape[1:3])} does not fit the cache {tuple(kv_cache.shape[1:3])}")
        kv_cache[:, :, :T] = r["qkv"][..., d:]
        return r["hidden"]

    @staticmethod
    def length_buckets(lens, max_buckets=16, bucket_cost=48):
        """Split sequences into <= ``max_buckets`` groups of similar length: minimises the padded positions
        sum(count_b * longest_b) plus ``bucket_cost`` positions per group.  Returns lists of indices, longest group first."""
        import numpy as np
        order = sorted(range(len(lens)), key=lambda i: -lens[i])
        n = len(order)
        if n == 0:
            return []
        Ls = np.asarray([lens[i] for i in order], dtype=np.float64)
        ii, jj = np.arange(n)[:, None], np.arange(1, n + 1)[None, :]
        W = np.where(ii < jj, (jj - ii) * Ls[:, None] + bucket_cost, np.inf)       # cost of the group order[i:j]
        prev = np.full(n + 1, np.inf)
        prev[0] = 0.0
        cuts, best_k, best_cost = [], 0, np.inf
        for k in range(1, min(max_buckets, n) + 1):                                 # best[k][j] = min_i best[k-1][i] + W[i][j]
            cand = prev[:n, None] + W
            cut = cand.argmin(axis=0)
            cur = np.concatenate(([np.inf], cand[cut, np.arange(n)]))
            cuts.append(cut)
            if cur[n] < best_cost:
                best_k, best_cost = k, cur[n]
            elif cur[n] > best_cost:
                break                                                               # one more group only adds its cost
            prev = cur
        groups, j, k = [], n, best_k
        while j > 0:
            i = int(cuts[k - 1][j - 1])
            groups.append(order[i:j])
            j, k = i, k - 1
        return groups[::-1]

    @torch.no_grad()
    def encode_groups(self, batches, embeds=False, want_hidden=True, want_qkv=False, want_meanpool=False):
        """Several right-padded batches -- ids [B_g, T_g] or, with ``embeds``, embeddings [B_g, T_g, d] -- in ONE launch
        sequence (``r4d_gpt2_encode_groups_ex_f32``): row-wise work runs over the concatenated rows, attention per batch.
        Returns dict(hidden [rows, d], qkv [L, rows, 3d], meanpool [sum B, d], row0 = first row of every batch)."""
        if not batches:
            raise ValueError("encode_groups: no batches")
        d, L = self.config.n_embd, self.config.n_layer
        ts = []
        for b in batches:
            if not b.is_cuda:
                raise _lib.R4DError("rag4dyg_amd runs on the GPU only: move inputs to 'cuda' (no CPU fallback)")
            ts.append(b.to(torch.float32).contiguous() if embeds else b.view(-1, b.shape[-1]).to(torch.int64).contiguous())
        n, dev = len(ts), ts[0].device
        lib = _lib.load()
        ops.range_flag(dev)
        c, w, _keep = self._c_structs()
        Bs = (ctypes.c_int32 * n)(*[int(t.shape[0]) for t in ts])
        Ts = (ctypes.c_int32 * n)(*[int(t.shape[1]) for t in ts])
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        row0, rows = [], 0
        for t in ts:
            row0.append(rows)
            rows += int(t.shape[0]) * int(t.shape[1])
        ws = ops.workspace(lib.r4d_gpt2_groups_workspace_bytes(ctypes.byref(c), n, Bs, Ts), dev, "gpt2")
        hidden = torch.empty(rows, d, dtype=torch.float32, device=dev) if want_hidden else None
        pool = torch.empty(sum(int(t.shape[0]) for t in ts), d, dtype=torch.float32, device=dev) if want_meanpool else None
        qkv = torch.empty(L, rows, 3 * d, dtype=torch.float32, device=dev) if want_qkv else None

        def ptr(t):
            return t.data_ptr() if t is not None else None
        _lib.check(lib.r4d_gpt2_encode_groups_ex_f32(ctypes.byref(c), ctypes.byref(w), n, None if embeds else ptrs,
                                                     ptrs if embeds else None, Bs, Ts, ptr(hidden), ptr(pool), ptr(qkv),
                                                     ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream),
                   "gpt2_encode_groups_ex")
        return dict(hidden=hidden, qkv=qkv, meanpool=pool, row0=row0)

    @torch.no_grad()
    def prefill_last(self, kv_cache, lens, input_ids=None, inputs_embeds=None, max_buckets=16):
        """``prefill`` for a RAGGED right-padded batch: the sequences are grouped by length (``length_buckets``), every
        group is padded to its own longest member only, and all groups run in one launch sequence (``encode_groups``) -- a
        causal model never lets padding reach a real position, so the cache rows [0, lens[i]) and the returned last real
        hidden row of every sequence, [B, d], are those of one padded forward, for a fraction of the positions
        (UCI-shaped prompts: 10.4 k padded positions -> 3.3 k)."""
        src = input_ids if input_ids is not None else inputs_embeds
        B, d, dev = src.shape[0], self.config.n_embd, src.device
        if len(lens) != B or min(lens) < 1 or max(lens) > src.shape[1]:
            raise ValueError("prefill_last: lens must hold one length in [1, T] per sequence")
        if max(lens) > kv_cache.shape[2] or B != kv_cache.shape[1]:
            raise ValueError(f"prefill_last: batch ({B}, {max(lens)}) does not fit the cache {tuple(kv_cache.shape[1:3])}")
        groups = self.length_buckets(list(lens), max_buckets)
        idxs = [torch.tensor(g, dtype=torch.long, device=dev) for g in groups]
        tbs = [max(lens[i] for i in g) for g in groups]
        # (range-guarded: the greedy loops take an argmax of whatever the logits hold -- a NaN prefill must not become token ids)
        r = self._range_guarded(dev, "prefill_last", lambda: self.encode_groups([src[ix, :tb] for ix, tb in zip(idxs, tbs)],
                                                                               embeds=input_ids is None, want_qkv=True))
        last_rows = [0] * B
        for g, ix, tb, r0 in zip(groups, idxs, tbs, r["row0"]):
            kv_cache[:, ix, :tb] = r["qkv"][:, r0:r0 + len(g) * tb, d:].view(-1, len(g), tb, 2 * d)
            for j, i in enumerate(g):
                last_rows[i] = r0 + j * tb + lens[i] - 1
        return r["hidden"][torch.tensor(last_rows, dtype=torch.long, device=dev)]

    @torch.no_grad()
    def decode_step(self, kv_cache, pos, input_ids=None, inputs_embeds=None):
        """One new position per sequence (``r4d_gpt2_decode_step_f32``): ``pos`` int32 [B] = positions already cached;
        row ``pos`` of the cache is written.  Returns the ln_f hidden row of the new position, [B, d]."""
        if (input_ids is None) == (inputs_embeds is None):
            raise ValueError("decode_step: specify exactly one of input_ids and inputs_embeds")
        L, B, t_cap, d2 = kv_cache.shape
        d = self.config.n_embd
        dev = kv_cache.device
        if not kv_cache.is_cuda or kv_cache.dtype != torch.float32 or not kv_cache.is_contiguous() or d2 != 2 * d \
                or L != self.config.n_layer:
            raise _lib.R4DError("decode_step: kv_cache must be a contiguous fp32 GPU tensor [n_layer, B, t_cap, 2d]")
        pos = pos.to(device=dev, dtype=torch.int32).contiguous()
        if input_ids is not None:
            input_ids = input_ids.to(device=dev, dtype=torch.int64).contiguous().view(B)
        else:
            inputs_embeds = inputs_embeds.to(device=dev, dtype=torch.float32).contiguous().view(B, d)
        lib = _lib.load()
        c, w, _keep = self._c_structs(decode=True)
        ws = ops.workspace(lib.r4d_gpt2_decode_workspace_bytes(ctypes.byref(c), B), dev, "gpt2_decode")
        hidden = torch.empty(B, d, dtype=torch.float32, device=dev)
        _lib.check(lib.r4d_gpt2_decode_step_f32(ctypes.byref(c), ctypes.byref(w),
                                                input_ids.data_ptr() if input_ids is not None else None,
                                                inputs_embeds.data_ptr() if inputs_embeds is not None else None,
                                                pos.data_ptr(), kv_cache.data_ptr(), B, t_cap, hidden.data_ptr(),
                                                ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream),
                   "gpt2_decode_step")
        return hidden

    def greedy_decoder(self, B, t_cap, slot=0):
        """A ``GreedyDecoder`` with room for ``B`` sequences of ``t_cap`` positions, reused across batches (its captured
        graph and buffers are keyed by (B, rounded t_cap)).  ``slot``: independent decoders of the same shape (one batch
        decodes while the next is prefilled into the other)."""
        t_cap = (int(t_cap) + 127) // 128 * 128
        cache = self.__dict__.setdefault("_greedy_decoders", {})
        fits = [k for k in cache if k[0] == B and k[1] >= t_cap and k[2] == slot]
        if fits:
            return cache[min(fits)]
        for k in [k for k in cache if k[0] == B and k[2] == slot]:           # a larger cache replaces the smaller ones
            cache.pop(k).close()
        dec = cache[(B, t_cap, slot)] = GreedyDecoder(self, B, t_cap)
        return dec

    @torch.no_grad()
    def encode_groups_meanpool(self, batches):
        """Mean-pooled embeddings of several right-padded batches in one fused launch sequence
        (``r4d_gpt2_encode_groups_f32``): values identical to encoding each batch on its own."""
        if not batches:
            raise ValueError("encode_groups_meanpool: no batches")
        ids = []
        for b in batches:
            if not b.is_cuda:
                raise _lib.R4DError("rag4dyg_amd runs on the GPU only: move inputs to 'cuda' (no CPU fallback)")
            ids.append(b.view(-1, b.shape[-1]).to(torch.int64).contiguous())
        n = len(ids)
        dev = ids[0].device
        lib = _lib.load()
        ops.range_flag(dev)
        c, w, _keep = self._c_structs()
        Bs = (ctypes.c_int32 * n)(*[int(t.shape[0]) for t in ids])
        Ts = (ctypes.c_int32 * n)(*[int(t.shape[1]) for t in ids])
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ids])
        ws = ops.workspace(lib.r4d_gpt2_groups_workspace_bytes(ctypes.byref(c), n, Bs, Ts), dev, "gpt2")
        out = torch.empty(sum(int(t.shape[0]) for t in ids), self.config.n_embd, dtype=torch.float32, device=dev)
        _lib.check(lib.r4d_gpt2_encode_groups_f32(ctypes.byref(c), ctypes.byref(w), n, ptrs, Bs, Ts, out.data_ptr(),
                                                  ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream),
                   "gpt2_encode_groups")
        return out

    def _presents(self, qkv):
        """``present = stack(k^T^T, v)`` per layer, [2,B,H,T,hd] (``modeling_gpt2.py:187``), as views of c_attn output."""
        L, B, T, d3 = qkv.shape
        d, H = d3 // 3, self.config.n_head
        k = qkv[..., d:2 * d].view(L, B, T, H, d // H).permute(0, 1, 3, 2, 4)
        v = qkv[..., 2 * d:].view(L, B, T, H, d // H).permute(0, 1, 3, 2, 4)
        return tuple(torch.stack((k[i], v[i])) for i in range(L))

    def forward(self, input_ids=None, past=None, attention_mask=None, token_type_ids=None, position_ids=None,
                head_mask=None, inputs_embeds=None):
        """Same outputs as ``modeling_gpt2.py:357-509``: (last_hidden_state, presents?, all_hidden_states?)."""
        for name, val in (("past", past), ("attention_mask", attention_mask), ("token_type_ids", token_type_ids),
                          ("position_ids", position_ids), ("head_mask", head_mask)):
            if val is not None:
                raise NotImplementedError(f"{name} is never passed on the encode-and-retrieve path "
                                          "(train_retriever.py:419,430; utils/model.py:222) and is not built")
        src = input_ids if input_ids is not None else inputs_embeds

        def run():
            return self.encode(input_ids, inputs_embeds, want_hidden=True, want_layers=self.output_hidden_states, want_qkv=self.output_past)
        # range guard (include/r4d.h, ABI v6): the reference-shaped call hands tensors to arbitrary code, so it reads the word itself
        r = self._range_guarded(src.device, "forward", run) if (src is not None and src.is_cuda) else run()
        outputs = (r["hidden"],)
        if self.output_past:
            outputs = outputs + (self._presents(r["qkv"]),)
        if self.output_hidden_states:
            outputs = outputs + (tuple(r["layers"][i] for i in range(self.config.n_layer)) + (r["hidden"],),)
        return outputs


class GreedyDecoder:
    """Greedy decoding with the loop state on the device (``r4d_gpt2_greedy_step_f32``): logits of the newest position,
    argmax, the stop rules of the reference's loops (``Evaluation_SimpleDyG.py:126-145``,
    ``Evaluation_generator.py:153-175``) and one key/value-cached step, per token, WITHOUT a host round trip -- replayed as
    a captured HIP graph (``r4d_gpt2_greedy_graph_create``; ``R4D_DECODE_GRAPH=0`` launches the same kernels one by
    one).  Owns the key/value cache ``self.cache`` [n_layer, B, t_cap, 2d]: prefill into it, then ``run``."""

    def __init__(self, transformer, B, t_cap):
        self.tr = transformer
        dev = transformer.wte.weight.device
        cfg = transformer.config
        d, V = cfg.n_embd, transformer.wte.num_embeddings
        self.B, self.t_cap, self.device = B, t_cap, dev
        self.cache = transformer.new_kv_cache(B, t_cap, dev)
        i32 = dict(dtype=torch.int32, device=dev)
        self.last = torch.zeros(B, d, dtype=torch.float32, device=dev)
        self.logits = torch.empty(B, V, dtype=torch.float32, device=dev)
        self.next = torch.zeros(B, dtype=torch.int64, device=dev)
        self.lens, self.pos = torch.zeros(B, **i32), torch.zeros(B, **i32)
        self.active, self.gen_len = torch.zeros(B, **i32), torch.zeros(B, **i32)
        self.out = torch.zeros(B, t_cap, **i32)
        self.params = torch.zeros(8, **i32)
        lib = _lib.load()
        c, _, _ = transformer._c_structs()
        self.ws = torch.empty(max(int(lib.r4d_gpt2_greedy_workspace_bytes(ctypes.byref(c), B)), 256), dtype=torch.uint8,
                              device=dev)
        self.state = _lib.GreedyStateC(self.last.data_ptr(), self.logits.data_ptr(), self.next.data_ptr(),
                                       self.lens.data_ptr(), self.pos.data_ptr(), self.active.data_ptr(),
                                       self.gen_len.data_ptr(), self.out.data_ptr(), self.params.data_ptr(), t_cap)
        self._graph = None
        self._graph_key = None
        self.graph_builds = 0
        self.use_graph = os.environ.get("R4D_DECODE_GRAPH", "1") != "0"

    def close(self):
        if self._graph is not None:
            _lib.load().r4d_decode_graph_destroy(self._graph)
            self._graph = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _steps(self, n):
        lib = _lib.load()
        c, w, layers = self.tr._c_structs(decode=True)
        stream = torch.cuda.current_stream().cuda_stream
        if self.use_graph:
            key = bytes(layers) + bytes((ctypes.c_void_p * 5)(w.wte, w.wpe, w.ln_f_w, w.ln_f_b, w.lm_head))   # every weight pointer baked in
            if self._graph is None or key != self._graph_key:
                self.close()
                g = ctypes.c_void_p()
                _lib.check(lib.r4d_gpt2_greedy_graph_create(ctypes.byref(c), ctypes.byref(w), ctypes.byref(self.state),
                                                            self.cache.data_ptr(), self.B, self.t_cap, self.ws.data_ptr(),
                                                            self.ws.numel(), ctypes.byref(g)), "gpt2_greedy_graph_create")
                self._graph, self._graph_key = g, key
                self.graph_builds += 1
            _lib.check(lib.r4d_decode_graph_launch(self._graph, n, stream), "decode_graph_launch")
        else:
            for _ in range(n):
                _lib.check(lib.r4d_gpt2_greedy_step_f32(ctypes.byref(c), ctypes.byref(w), ctypes.byref(self.state),
                                                        self.cache.data_ptr(), self.B, self.t_cap, self.ws.data_ptr(),
                                                        self.ws.numel(), stream), "gpt2_greedy_step")

    @torch.no_grad()
    def run(self, last, lens, max_gen, len_limit, eos=(), poll=8):
        """Generate from the prefilled cache.  ``last`` [B,d]: ln_f row of every sequence's last prompt position;
        ``lens`` [B]: positions cached.  A sequence stops after ``max_gen`` tokens, after any id in ``eos``, or once its
        length reaches ``len_limit``.  Returns the generated ids per sequence (lists of ints)."""
        eos = [int(e) for e in eos][:4]
        self.last.copy_(last)
        self.lens.copy_(lens.to(torch.int32))
        self.active.fill_(1)
        self.gen_len.zero_()
        self.params.copy_(torch.tensor([int(max_gen), int(min(len_limit, self.t_cap)), len(eos)] + eos + [0] * (5 - len(eos)),
                                       dtype=torch.int32))
        n, done = (int(max_gen) if max_gen <= 16 else poll), 0
        while True:
            self._steps(n)
            done += n
            if done > self.t_cap or not bool(self.active.any()):    # one host sync per `poll` tokens
                break
            n = poll
        out, g = self.out.cpu(), self.gen_len.cpu()
        return [out[i, :int(g[i])].tolist() for i in range(self.B)]


class _LMHeadBase(_PreTrained):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.transformer = GPT2Model(config)
        self.lm_head = nn.Linear(config.n_embd, config.vocab_size, bias=False)
        self.apply(self._init_weights)
        self.tie_weights()

    def tie_weights(self):
        """``modeling_utils.py:155-181``: lm_head.weight is the wte Parameter."""
        self.lm_head.weight = self.transformer.wte.weight
        self._untied_by_checkpoint = False

    def get_output_embeddings(self):
        return self.lm_head

    def lm_head_is_tied(self):
        return self.lm_head.weight is self.transformer.wte.weight

    def _link_transformer(self):
        """The decode kernels read lm_head through the transformer's weight struct: the transformer gets a getter that
        follows whatever Parameter THIS model's lm_head.weight currently is (tied, untied, re-tied).  Held through a weak
        reference (no cycle) and rebuilt for every copy / unpickled model (``__setstate__``): a deep copy must resolve its
        OWN lm_head, not the original's."""
        tr = self._modules.get("transformer")
        if isinstance(tr, GPT2Model):
            ref = weakref.ref(self)

            def head():
                owner = ref()
                return owner.lm_head.weight if owner is not None and "lm_head" in owner._modules else None
            tr.__dict__["_lm_head_weight"] = head

    def __setattr__(self, name, value):
        super().__setattr__(name, value)
        if name == "transformer" and isinstance(value, GPT2Model):
            self._link_transformer()

    def __setstate__(self, state):
        super().__setstate__(state)
        self._link_transformer()

    def load_state_dict(self, state_dict, strict=True, **kw):
        """``nn.Module.load_state_dict`` with the reference's UNTIED checkpoints handled.  The reference unties lm_head from
        wte whenever it replaces ``transformer.wte`` or ``model.transformer`` without re-tying (hepth node features,
        ``utils/tokenizer.py:56-66``; ``load_and_freeze_params``, ``utils/model.py:71-78``), trains the two separately and
        saves both.  Loading such a checkpoint into a TIED model would copy ``transformer.wte.weight`` and then
        ``lm_head.weight`` into the same Parameter (the input embedding silently becomes the lm_head).  Here the model is
        untied first, so both tensors load as saved -- the state the reference's own training process evaluates with
        (``--evaluate_during_training``).  ``R4D_REFERENCE_EVAL_TIE=1`` keeps the tie, reproducing what the reference's
        eval-only run (``main_generator.py:108-121`` on a freshly built, tied model) computes."""
        import os as _os
        a, b = state_dict.get("transformer.wte.weight"), state_dict.get("lm_head.weight")
        if a is not None and b is not None and self.lm_head_is_tied() and _os.environ.get("R4D_REFERENCE_EVAL_TIE") != "1" \
                and (a.shape != b.shape or not torch.equal(a, b)):
            w = self.transformer.wte.weight
            self.lm_head.weight = nn.Parameter(w.detach().clone())
            self._untied_by_checkpoint = True
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def resize_token_embeddings(self, new_num_tokens=None):
        emb = self.transformer.resize_token_embeddings(new_num_tokens)
        self.lm_head = nn.Linear(self.config.n_embd, emb.num_embeddings, bias=False).to(emb.weight.device)
        self.config.vocab_size = emb.num_embeddings
        self.tie_weights()
        return emb

    def _lm(self, input_ids, labels, inputs_embeds, **unsupported):
        tr = self.transformer(input_ids, inputs_embeds=inputs_embeds, **unsupported)
        hidden = tr[0]
        lm_logits = ops.lm_logits(hidden, self.lm_head.weight)                 # == wte unless the checkpoint untied them
        outputs = (lm_logits,) + tr[1:]
        if labels is not None:                     # shifted CE, modeling_gpt2.py:604-615 (metric only; torch op on device)
            shift_logits = lm_logits[..., :-1, :].contiguous()
            shift_labels = labels[..., 1:].contiguous()
            loss = nn.functional.cross_entropy(shift_logits.view(-1, shift_logits.size(-1)), shift_labels.view(-1))
            outputs = (loss,) + outputs
        return outputs, hidden


class GPT2LMHeadModel(_LMHeadBase):
    """SimpleDyG flavour, ``models/modeling_gpt2.py:517-617``: returns ``(loss?), lm_logits, presents, ...``."""

    def forward(self, input_ids=None, past=None, attention_mask=None, token_type_ids=None, position_ids=None,
                head_mask=None, inputs_embeds=None, labels=None):
        outputs, _ = self._lm(input_ids, labels, inputs_embeds, past=past, attention_mask=attention_mask,
                              token_type_ids=token_type_ids, position_ids=position_ids, head_mask=head_mask)
        return outputs


class GPT2LMHeadModelRAG(_LMHeadBase):
    """Retriever / generator flavour, ``models/modeling_rag.py:569-687``: returns ``(outputs, hidden_states)``.

    ``encode_meanpool`` is the retrieve-path entry: it skips the lm_head GEMM the retriever computes and throws
    away (``train_retriever.py:419-420``; 31-48 % of the reference forward).
    """

    def __init__(self, config):
        super().__init__(config)
        self.mlp_fusion = None                       # models/modeling_rag.py:582-584: created on demand, so that a
        self.gnn_fusion = None                       # generator checkpoint's ``{mlp,gnn}_fusion.*`` keys load

    def get_mlp(self, input_size, output_size, n_layers=1):
        """``modeling_rag.py:589-593``."""
        if self.mlp_fusion is None:
            from .generator import MLP_custom
            self.mlp_fusion = MLP_custom(input_size, output_size, n_layers)
        return self.mlp_fusion

    def get_gnn(self, nfeat, nhid, nclass, n_layers, dropout):
        """``modeling_rag.py:595-600``."""
        if self.gnn_fusion is None:
            from .generator import GNN
            self.gnn_fusion = GNN(nfeat, nhid, nclass, n_layers, dropout)
        return self.gnn_fusion

    def forward(self, input_ids=None, past=None, attention_mask=None, token_type_ids=None, position_ids=None,
                head_mask=None, inputs_embeds=None, labels=None):
        return self._lm(input_ids, labels, inputs_embeds, past=past, attention_mask=attention_mask,
                        token_type_ids=token_type_ids, position_ids=position_ids, head_mask=head_mask)

    @torch.no_grad()
    def encode_groups_meanpool(self, batches):
        """Several reference batches in one fused launch sequence; same values as ``encode_meanpool`` per batch."""
        return self.transformer.encode_groups_meanpool(batches)

    @torch.no_grad()
    def encode_meanpool(self, input_ids):
        """``_, h = model(input_ids); torch.mean(h, dim=1)`` (``train_retriever.py:419-420``) fused on device."""
        return self.transformer.encode(input_ids, want_hidden=False, want_meanpool=True)["meanpool"]
