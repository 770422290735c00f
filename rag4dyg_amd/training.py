"""Retriever TRAINING: the data, the augmentation, one ``train_epoch`` iteration on the gfx950 kernels
(five encoder forwards with saved activations, the two contrastive losses, the encoder's backward pass, gradient clipping and
the AdamW update) and the training loop of ``main_retriever.py --do_train``.

Mirrors ``dataloader/retriever.py:68-111`` (``PairSequenceDataset``), ``models/modeling_rag.py:774-840`` (``_aug``),
``train/train_retriever.py:40-98,120-354`` (the two contrastive losses -- on the device, ``csrc/losses.hip`` --, ``adjust_learning_rate``,
``train_epoch``, ``train``) and ``utils/model.py:56-102`` (checkpoints, the ``transformers.AdamW`` configuration).  The five
batches of a step (anchor, positive, hard negative, two augmented views) go through ONE launch sequence over their concatenated
rows (``r4d_gpt2_train_forward_f32`` / ``_backward_f32``); the losses themselves are [B, 3B] / [2B, 2B] similarity tables:
torch ops (and torch autograd) on the [5, B, d] embeddings only.  In ``model.eval()`` terms the forward, the losses and every
parameter gradient equal the reference's (tests/golden/g8); in training mode dropout draws its masks from the library's
counter-based generator (the reference's torch RNG stream cannot be reproduced), checked against the oracle given the same masks.
"""
import math
import os
import random
import warnings

import numpy as np
import torch

from .dataloader import read_nonblank_lines


class PairSequenceDataset(torch.utils.data.Dataset):
    """``dataloader/retriever.py:68-111``: (anchor, positive, negative) index triples of ``train_index.retrieval`` resolved
    against the history part of the training lines."""

    def __init__(self, tokenizer, args, file_path, block_size=512):
        assert os.path.isfile(file_path)
        train_lines = [line.split('<|pre|>')[0].strip() for line in read_nonblank_lines(args.train_data_file)]
        triples = [list(map(int, line.split())) for line in read_nonblank_lines(file_path)]
        examples = tokenizer(train_lines, add_special_tokens=True, max_length=block_size)["input_ids"]
        self.anchor, self.positive, self.negative = [], [], []
        self.anchor_idx, self.positive_idx, self.negative_idx = [], [], []
        for a, p, n in triples:
            self.anchor.append(examples[a]); self.positive.append(examples[p]); self.negative.append(examples[n])
            self.anchor_idx.append([a]); self.positive_idx.append([p]); self.negative_idx.append([n])

    def __len__(self):
        return len(self.anchor)

    def __getitem__(self, i):
        t = lambda x: torch.tensor(x, dtype=torch.long)
        return (t(self.anchor[i]), t(self.positive[i]), t(self.negative[i]), t(self.anchor_idx[i]), t(self.positive_idx[i]),
                t(self.negative_idx[i]))


def aug(batch_seqs, eta, gamma, mask_token):
    """``_aug`` (``models/modeling_rag.py:774-840``): view 1 = crop (``eta``), view 2 = mask (``gamma``), python ``random``
    as upstream.  Kept as is: a row's length is its number of non-zero ids; the crop lands at the end of an all-zero row;
    the mask view overwrites the row it was taken from (``seq[:]`` of an ndarray is a view)."""
    seqs = batch_seqs.tolist()
    lengths = batch_seqs.count_nonzero(dim=1).tolist()
    view1, view2 = [], []
    for seq, length in zip(seqs, lengths):
        seq = np.asarray(seq.copy(), dtype=np.int64)
        if length > 1:
            num_left = math.floor(length * eta)                              # item_crop
            crop_begin = random.randint(4, length - num_left)
            c = np.zeros_like(seq)
            if crop_begin != 0:
                c[-num_left:] = seq[-(crop_begin + num_left):-crop_begin]
            else:
                c[-num_left:] = seq[-(crop_begin + num_left):]
            view1.append(c.tolist() if num_left > 0 else seq.tolist())
            num_mask = math.floor(length * gamma)                            # item_mask
            mask_index = [-i - 1 for i in random.sample(range(length), k=num_mask)]
            seq[mask_index] = mask_token
            view2.append(seq.tolist())
        else:
            view1.append(seq.tolist()); view2.append(seq.tolist())
    dev = batch_seqs.device
    return torch.tensor(view1, dtype=torch.long, device=dev), torch.tensor(view2, dtype=torch.long, device=dev)


def retriever_losses(args, emb, t_anchor, t_pos, t_neg, grad_scale=1.0, want_grad=True):
    """The step's two contrastive losses on the device (``csrc/losses.hip``, ``r4d_retriever_losses_f32``):
    ``CLtime_loss`` (``train/train_retriever.py:40-72``) on emb[0..2] with the query times, ``alpha * info_nce`` (:84-98) on the
    two augmented views emb[3], emb[4], and -- with ``want_grad`` -- ``grad_scale * d(loss)/d(emb)``.  ``emb`` [5, B, d] fp32;
    the times are the ``all_query_time[idx]`` tensors of the reference ([B] or [B, 1]).  Returns (losses [3] = CLtime,
    alpha * info_nce, their sum; device tensor, no host sync), d_emb [5, B, d] or None).  The last, smaller batch of an epoch
    needs no rebuilt mask: info_nce's negatives are simply "every other row but the partner" for whatever B arrives."""
    lib = _lib.load()
    emb = emb.contiguous()
    _five, B, d = emb.shape
    dev = emb.device
    ts = [t.to(device=dev, dtype=torch.float32).reshape(-1).contiguous() for t in (t_anchor, t_pos, t_neg)]
    if any(t.numel() != B for t in ts):
        raise _lib.R4DError(f"retriever_losses: {B} sequences but times of {[t.numel() for t in ts]} elements")
    ws = ops.workspace(lib.r4d_retriever_losses_workspace_bytes(B), dev, "losses")
    losses = torch.empty(3, dtype=torch.float32, device=dev)
    demb = torch.empty_like(emb) if want_grad else None
    _lib.check(lib.r4d_retriever_losses_f32(emb.data_ptr(), ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr(), B, d,
                                            float(args.temperature), float(args.lambda_decay), float(args.alpha), float(grad_scale),
                                            losses.data_ptr(), demb.data_ptr() if want_grad else None, ws.data_ptr(), ws.numel(),
                                            torch.cuda.current_stream().cuda_stream), "retriever_losses")
    return losses, demb


@torch.no_grad()
def training_step_forward(args, model, batch, all_query_time):
    """Forward half of one ``train_epoch`` iteration (``train/train_retriever.py:164-196``): ``batch`` = (anchor, positive,
    negative, anchor_idx, positive_idx, negative_idx) as the ``PairSequenceDataset`` loader yields them.  Returns
    dict(cl_loss, aug_loss, loss, embeddings [5, B, d], aug1, aug2)."""
    anchor_seq, pos_seq, neg_seq, anchor_idx, pos_idx, neg_idx = batch
    dev = args.device
    anchor_seq, pos_seq, neg_seq = anchor_seq.to(dev), pos_seq.to(dev), neg_seq.to(dev)
    aug1, aug2 = aug(anchor_seq, model.config.eta, model.config.gamma, model.config.vocab_size - 1)
    B = anchor_seq.size(0)
    emb = model.encode_groups_meanpool([anchor_seq, pos_seq, neg_seq, aug1, aug2]).view(5, B, -1)   # one fused launch sequence
    t = all_query_time
    losses, _ = retriever_losses(args, emb, t[anchor_idx], t[pos_idx], t[neg_idx], want_grad=False)
    return dict(cl_loss=losses[0], aug_loss=losses[1], loss=losses[2], embeddings=emb, aug1=aug1, aug2=aug2)


# ------------------------------------------------------------------------------------------------ backward + optimizer
import ctypes                                                                   # noqa: E402

from . import _lib, ops                                                         # noqa: E402
from .gpt2 import note_raw_parameter_write                                     # noqa: E402

_LAYER_PARAMS = (("ln_1_w", "ln_1.weight"), ("ln_1_b", "ln_1.bias"), ("c_attn_w", "attn.c_attn.weight"),
                 ("c_attn_b", "attn.c_attn.bias"), ("attn_proj_w", "attn.c_proj.weight"), ("attn_proj_b", "attn.c_proj.bias"),
                 ("ln_2_w", "ln_2.weight"), ("ln_2_b", "ln_2.bias"), ("c_fc_w", "mlp.c_fc.weight"), ("c_fc_b", "mlp.c_fc.bias"),
                 ("mlp_proj_w", "mlp.c_proj.weight"), ("mlp_proj_b", "mlp.c_proj.bias"))


class EncoderTrainer:
    """Forward-with-saved-activations and backward of the SimpleDyG encoder on the HIP kernels
    (``r4d_gpt2_train_forward_f32`` / ``r4d_gpt2_train_backward_f32``), for a ``GPT2LMHeadModelRAG`` whose parameters live
    on the GPU.  ``grads`` maps the reference's parameter names (``transformer.h.0.attn.c_attn.weight`` ...) to gradient
    tensors; ``lm_head.weight`` has none (the retriever discards the logits) unless it is the tied ``wte`` Parameter."""

    def __init__(self, model, dropout=None, seed=0):
        """``dropout``: None -> the model config's ``embd_pdrop`` / ``attn_pdrop`` / ``resid_pdrop`` when the module is in
        training mode (``model.train()``, ``train_retriever.py:161``), the identity in eval mode; or an explicit
        (embd_p, attn_p, resid_p).  ``seed`` keys the counter-based mask generator; every forward advances its step."""
        self.model = model
        self.dropout, self.seed, self.step = dropout, int(seed), 0
        self._drop_struct = None
        tr = model.transformer
        self.params = {"transformer.wte.weight": tr.wte.weight, "transformer.wpe.weight": tr.wpe.weight,
                       "transformer.ln_f.weight": tr.ln_f.weight, "transformer.ln_f.bias": tr.ln_f.bias}
        for i, blk in enumerate(tr.h):
            sd = dict(blk.named_parameters())
            for _f, name in _LAYER_PARAMS:
                self.params[f"transformer.h.{i}.{name}"] = sd[name]
        for n, p in self.params.items():
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.R4DError(f"{n}: training needs contiguous fp32 parameters on the GPU (no CPU fallback)")
        # every gradient is a view of ONE flat buffer (64-float aligned slices): the gradient norm is one launch over it, an
        # accumulation step one add, and a data-parallel step ONE all-reduce over RCCL instead of one per tensor
        offs, total = {}, 0
        for n, p in self.params.items():
            offs[n] = total
            total += (p.numel() + 63) // 64 * 64
        dev = tr.wte.weight.device
        self.flat_grads = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grads = {n: self.flat_grads[offs[n]:offs[n] + p.numel()].view_as(p) for n, p in self.params.items()}
        self.flat_accum = None
        self._ws = None
        self._saved = None
        # [out,in] copies of the four Conv1D weights of every block for the forward GEMMs (both operands k-contiguous: the
        # fast kernel); refreshed by ``refresh_transposed()`` after every parameter update
        self.use_wt = os.environ.get("R4D_TRAIN_WT", "1") != "0"
        self._wt = {}
        if self.use_wt:
            for i in range(len(tr.h)):
                for f in ("c_attn_w", "attn_proj_w", "c_fc_w", "mlp_proj_w"):
                    w = self.params[f"transformer.h.{i}.{dict(_LAYER_PARAMS)[f]}"]
                    self._wt[(i, f)] = torch.empty(w.shape[1], w.shape[0], dtype=torch.float32, device=dev)
        # bf16x3 planes of the same weights: [3,out,in] for the forward GEMMs, [3,in,out] for the data gradients dx = dy . W^T
        # (ops.split3_planes; the bf16 matrix cores at fp32 accuracy); refreshed with the copies above
        self.use_s3 = ops.gemm_split3_enabled() and os.environ.get("R4D_TRAIN_SPLIT3", "1") != "0"
        self._w3, self._w3t = {}, {}
        # f16x2 mode (round 5): f16x2 lines [out, in/32, 2, 32] of the same weights for the FORWARD GEMMs -- the fp16 matrix cores with
        # three products per fp32 product (csrc/gemm_h2.hip).  Every gradient GEMM keeps bf16x3: a gradient operand (1e-5 .. 1e-8) needs
        # fp32's exponent range (csrc/train.hip: bwd_data).  R4D_TRAIN_H2=0 keeps bf16x3 in the forward pass too.
        self.use_h2 = self.use_s3 and ops.gemm_mode() == "f16x2" and os.environ.get("R4D_TRAIN_H2", "1") != "0"
        self._h2 = {}
        if self.use_wt or self.use_s3:
            self.refresh_transposed()

    @torch.no_grad()
    def refresh_transposed(self):
        """Bring the [out,in] weight copies and the bf16x3 planes up to date (call after every optimizer step / load_state_dict)."""
        for (i, f), wt in self._wt.items():
            wt.copy_(self.params[f"transformer.h.{i}.{dict(_LAYER_PARAMS)[f]}"].t())
        if self.use_s3:
            lib = _lib.load()
            stream = torch.cuda.current_stream().cuda_stream
            for i in range(len(self.model.transformer.h)):
                for f in ("c_attn_w", "attn_proj_w", "c_fc_w", "mlp_proj_w"):
                    w = self.params[f"transformer.h.{i}.{dict(_LAYER_PARAMS)[f]}"]
                    K, N = w.shape
                    if K % 32 or N % 32:
                        continue
                    if (i, f) not in self._w3:
                        self._w3[(i, f)] = torch.empty(3, N, K, dtype=torch.int16, device=w.device)
                        self._w3t[(i, f)] = torch.empty(3, K, N, dtype=torch.int16, device=w.device)
                    _lib.check(lib.r4d_split3_planes_bf16(w.data_ptr(), K, N, 0, self._w3[(i, f)].data_ptr(), stream), "split3_planes")
                    # the data-gradient operand: W itself read as an [N' = K rows, K' = N contiguous] matrix
                    _lib.check(lib.r4d_split3_planes_bf16(w.data_ptr(), N, K, 1, self._w3t[(i, f)].data_ptr(), stream), "split3_planes")
                    if self.use_h2:
                        if (i, f) not in self._h2:
                            self._h2[(i, f)] = torch.empty(N, K // 32, 2, 32, dtype=torch.int16, device=w.device)
                        _lib.check(lib.r4d_split2_planes_f16(w.data_ptr(), K, N, 0, self._h2[(i, f)].data_ptr(), stream), "split2_planes")

    def _structs(self):
        tr = self.model.transformer
        cfg = tr.config
        c = _lib.GPT2ConfigC(cfg.n_layer, cfg.n_head, cfg.n_embd, tr.wte.num_embeddings, tr.wpe.num_embeddings,
                             cfg.layer_norm_epsilon)
        layers = (_lib.GPT2LayerC * cfg.n_layer)()
        glayers = (_lib.GPT2LayerGradsC * cfg.n_layer)()
        for i in range(cfg.n_layer):
            vals = [self.params[f"transformer.h.{i}.{name}"].data_ptr() for _f, name in _LAYER_PARAMS]
            wts = [self._wt[(i, f)].data_ptr() if self.use_wt else None for f in ("c_attn_w", "attn_proj_w", "c_fc_w", "mlp_proj_w")]
            fs = ("c_attn_w", "attn_proj_w", "c_fc_w", "mlp_proj_w")
            w3 = [self._w3[(i, f)].data_ptr() if (i, f) in self._w3 else None for f in fs]
            w3t = [self._w3t[(i, f)].data_ptr() if (i, f) in self._w3t else None for f in fs]
            layers[i] = _lib.GPT2LayerC(*vals, *wts, *w3, *w3t)                               # copies / planes kept current by refresh_transposed
            if self.use_h2 and ops.gemm_mode() == "f16x2":
                for f, name in zip(fs, ("c_attn", "attn_proj", "c_fc", "mlp_proj")):
                    if (i, f) in self._h2:
                        setattr(layers[i], name + "_h2", self._h2[(i, f)].data_ptr())
            glayers[i] = _lib.GPT2LayerGradsC(*[self.grads[f"transformer.h.{i}.{name}"].data_ptr() for _f, name in _LAYER_PARAMS])
        w = _lib.GPT2WeightsC(tr.wte.weight.data_ptr(), tr.wpe.weight.data_ptr(), tr.ln_f.weight.data_ptr(),
                              tr.ln_f.bias.data_ptr(), layers, None)
        g = _lib.GPT2GradsC(self.grads["transformer.wte.weight"].data_ptr(), self.grads["transformer.wpe.weight"].data_ptr(),
                            self.grads["transformer.ln_f.weight"].data_ptr(), self.grads["transformer.ln_f.bias"].data_ptr(), glayers)
        return c, w, g, (layers, glayers)

    def _dropout_struct(self):
        if self.dropout is not None:
            pe, pa, pr = self.dropout
        elif self.model.training:
            c = self.model.config
            pe, pa, pr = c.embd_pdrop, c.attn_pdrop, c.resid_pdrop
        else:
            return None
        if pe <= 0 and pa <= 0 and pr <= 0:
            return None
        return _lib.TrainDropoutC(float(pe), float(pa), float(pr), self.seed & (2 ** 64 - 1), self.step)

    @torch.no_grad()
    def forward(self, batches):
        """Mean-pooled embeddings [sum B, d] of the right-padded id batches; keeps the activations for ``backward``."""
        ids = [b.view(-1, b.shape[-1]).to(torch.int64).contiguous() for b in batches]
        n, dev = len(ids), ids[0].device
        lib = _lib.load()
        c, w, g, keep = self._structs()
        Bs = (ctypes.c_int32 * n)(*[int(t.shape[0]) for t in ids])
        Ts = (ctypes.c_int32 * n)(*[int(t.shape[1]) for t in ids])
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ids])
        nbytes = lib.r4d_gpt2_train_workspace_bytes(ctypes.byref(c), n, Bs, Ts)
        if nbytes == 0:
            raise _lib.R4DError("gpt2 train: bad batch shapes")
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        out = torch.empty(sum(int(t.shape[0]) for t in ids), self.model.config.n_embd, dtype=torch.float32, device=dev)
        self.step += 1
        self._drop_struct = self._dropout_struct()                 # the backward of this step regenerates the same masks
        _lib.check(lib.r4d_gpt2_train_forward_f32(ctypes.byref(c), ctypes.byref(w), n, ptrs, Bs, Ts, out.data_ptr(),
                                                  ctypes.byref(self._drop_struct) if self._drop_struct is not None else None,
                                                  self._ws.data_ptr(), self._ws.numel(), torch.cuda.current_stream().cuda_stream),
                   "gpt2_train_forward")
        self._saved = (ids, n, Bs, Ts, ptrs)
        return out

    @torch.no_grad()
    def backward(self, d_embeddings):
        """dLoss/d(embeddings) [sum B, d] -> ``self.grads`` (overwritten)."""
        if self._saved is None:
            raise RuntimeError("backward() before forward()")
        ids, n, Bs, Ts, ptrs = self._saved
        lib = _lib.load()
        c, w, g, keep = self._structs()
        de = d_embeddings.to(torch.float32).contiguous()
        _lib.check(lib.r4d_gpt2_train_backward_f32(ctypes.byref(c), ctypes.byref(w), ctypes.byref(g), n, ptrs, Bs, Ts, de.data_ptr(),
                                                   ctypes.byref(self._drop_struct) if self._drop_struct is not None else None,
                                                   self._ws.data_ptr(), self._ws.numel(), torch.cuda.current_stream().cuda_stream),
                   "gpt2_train_backward")
        self._saved = None
        return self.grads

    @torch.no_grad()
    def accumulate(self):
        """``loss.backward()`` of a micro-step under ``--gradient_accumulation_steps`` > 1 (``train_retriever.py:202-212``): the
        gradients just computed are added to the running sum."""
        if self.flat_accum is None:
            self.flat_accum = torch.zeros_like(self.flat_grads)
        self.flat_accum.add_(self.flat_grads)

    @torch.no_grad()
    def take_accumulated(self):
        """The running sum becomes the gradient the optimizer sees; the sum restarts at zero (``model.zero_grad()``)."""
        if self.flat_accum is not None:
            self.flat_grads.copy_(self.flat_accum)
            self.flat_accum.zero_()

    @torch.no_grad()
    def all_reduce_mean(self):
        """Data-parallel step (``DistributedDataParallel``, ``train_retriever.py:260-266``): gradients averaged over the ranks --
        one collective over the flat buffer."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat_grads)
            self.flat_grads.div_(dist.get_world_size())


class AdamW:
    """``transformers.AdamW`` as ``utils/model.py:80-93`` configures it -- bias correction on, decoupled weight decay applied
    after the update, no decay for names containing ``bias`` / ``LayerNorm.weight`` (the GPT-2 LayerNorms are called ln_1 /
    ln_2 / ln_f, so their gains DO decay: upstream quirk kept) -- with ``clip_grad_norm_`` folded in: the squared norm of
    all gradients is accumulated on the device and every update kernel scales its gradient by the clip coefficient."""

    def __init__(self, params, grads, lr, eps=1e-8, weight_decay=0.0, betas=(0.9, 0.999), flat_grads=None):
        self.params, self.grads = params, grads
        self.flat_grads = flat_grads                 # EncoderTrainer.flat_grads: the norm is then one launch (padding is zero)
        self.lr, self.eps, self.betas = lr, eps, betas
        no_decay = ("bias", "LayerNorm.weight")
        self.wd = {n: (0.0 if any(nd in n for nd in no_decay) else weight_decay) for n in params}
        self.m = {n: torch.zeros_like(p) for n, p in params.items()}
        self.v = {n: torch.zeros_like(p) for n, p in params.items()}
        self.t = 0
        dev = next(iter(params.values())).device
        self.sumsq = torch.zeros(1025, dtype=torch.float32, device=dev)        # R4D_SUMSQ_FLOATS: [0] total, rest scratch

    @torch.no_grad()
    def step(self, max_grad_norm=0.0):
        lib = _lib.load()
        stream = torch.cuda.current_stream().cuda_stream
        self.t += 1
        self.sumsq.zero_()
        if max_grad_norm and max_grad_norm > 0:
            for g in ([self.flat_grads] if self.flat_grads is not None else self.grads.values()):
                _lib.check(lib.r4d_sumsq_accumulate_f32(g.data_ptr(), g.numel(), self.sumsq.data_ptr(), stream), "sumsq")
        for n, p in self.params.items():
            _lib.check(lib.r4d_adamw_step_f32(p.data_ptr(), self.grads[n].data_ptr(), self.m[n].data_ptr(), self.v[n].data_ptr(),
                                              p.numel(), float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                              float(self.wd[n]), self.t, self.sumsq.data_ptr() if max_grad_norm else None,
                                              float(max_grad_norm or 0.0), stream), "adamw")
        note_raw_parameter_write()        # the kernels wrote the parameters behind torch's version counters: derived copies are stale

    def grad_norm(self):
        return float(self.sumsq[0].sqrt().item())

    def load_state(self, state):
        """Moments, step count and learning rate written by :func:`save_checkpoint` (``r4d_optimizer.pt``): what
        ``get_optimizer_scheduler`` does with ``optimizer.pt`` / ``scheduler.pt`` when training continues from a checkpoint
        directory (``utils/model.py:96-102``)."""
        if state.get("format") != "rag4dyg_amd.AdamW" or set(state["m"]) != set(self.m):
            raise _lib.R4DError("AdamW.load_state: not an r4d_optimizer.pt of this model")
        for n in self.m:
            if state["m"][n].shape != self.m[n].shape:
                raise _lib.R4DError(f"AdamW.load_state: {n}: shape {tuple(state['m'][n].shape)} != {tuple(self.m[n].shape)}")
            self.m[n].copy_(state["m"][n])
            self.v[n].copy_(state["v"][n])
        self.t, self.lr = int(state["t"]), float(state["lr"])


def _to_device(x, dev):
    """Host tensor -> device through pinned memory, asynchronously: a pageable ``.to(device)`` is a blocking copy that first
    waits for everything queued on the stream, i.e. for the previous training step."""
    if x.is_cuda or torch.device(dev).type != "cuda":
        return x.to(dev)
    return x.pin_memory().to(dev, non_blocking=True)


def training_step(args, model, trainer, optimizer, batch, all_query_time, micro_step=0, sync=True):
    """One iteration of ``train_epoch`` (``train/train_retriever.py:164-214``) on the device: five forwards (one launch
    sequence), the two contrastive losses and their gradient on the [5, B, d] embeddings (``retriever_losses``: three small
    launches, no framework autograd anywhere in the step), the encoder's backward pass
    and -- every ``gradient_accumulation_steps``-th micro-step -- the gradient average over the data-parallel ranks,
    gradient clipping and the AdamW update.  Returns dict(loss, cl_loss, aug_loss, stepped); with ``sync=False`` the three
    losses are 0-d device tensors instead of floats, so the host does not wait for the GPU and prepares the next batch (the
    python ``random`` augmentation, ~10 ms per 64 sequences) while this one computes."""
    anchor_seq, pos_seq, neg_seq, anchor_idx, pos_idx, neg_idx = batch
    dev = args.device
    # augmentation on the loader's HOST copy: no device-to-host round trip (and no wait for the previous step) before the launch
    aug1, aug2 = aug(anchor_seq, model.config.eta, model.config.gamma, model.config.vocab_size - 1)
    anchor_seq, pos_seq, neg_seq, aug1, aug2 = (_to_device(x, dev) for x in (anchor_seq, pos_seq, neg_seq, aug1, aug2))
    if all_query_time.device == anchor_seq.device:              # times resident on the GPU (train()): index them there
        anchor_idx, pos_idx, neg_idx = (_to_device(x, dev) for x in (anchor_idx, pos_idx, neg_idx))
    B = anchor_seq.size(0)
    emb = trainer.forward([anchor_seq, pos_seq, neg_seq, aug1, aug2])
    t = all_query_time
    gas = max(1, int(getattr(args, "gradient_accumulation_steps", 1)))
    losses, demb = retriever_losses(args, emb.view(5, B, -1), t[anchor_idx], t[pos_idx], t[neg_idx], grad_scale=1.0 / gas)
    cl, au, loss = losses[0], losses[1], losses[2] / gas
    trainer.backward(demb.view(5 * B, -1))
    if gas > 1:
        trainer.accumulate()
    stepped = (micro_step + 1) % gas == 0
    if stepped:
        if gas > 1:
            trainer.take_accumulated()
        trainer.all_reduce_mean()
        optimizer.step(getattr(args, "max_grad_norm", 0.0))
        trainer.refresh_transposed()
        for c in ("_wt_cache", "_w3_cache", "_h2_cache", "_fold_cache"):       # the inference path's derived weights are stale now (also
            model.transformer.__dict__.pop(c, None)               # guarded by gpt2.note_raw_parameter_write in AdamW.step)
    if not sync:
        return dict(loss=loss, cl_loss=cl, aug_loss=au, stepped=stepped)
    return dict(loss=float(loss.item()), cl_loss=float(cl.item()), aug_loss=float(au.item()), stepped=stepped)


# ------------------------------------------------------------------------------------------------ training loop
def adjust_learning_rate(args, optimizer, epoch, base_lr, i, iteration_per_epoch):
    """``train/train_retriever.py:120-130``: linear warm-up over ``warmup_steps`` EPOCHS, then a half cosine."""
    T = epoch * iteration_per_epoch + i
    warmup_iters = args.warmup_steps * iteration_per_epoch
    total_iters = (args.num_train_epochs - args.warmup_steps) * iteration_per_epoch
    if epoch < args.warmup_steps:
        lr = base_lr * 1.0 * T / warmup_iters
    else:
        lr = 0.5 * base_lr * (1 + math.cos(1.0 * (T - warmup_iters) / total_iters * math.pi))
    optimizer.lr = lr


def save_checkpoint(model, optimizer, tokenizer, args, global_step):
    """``utils/model.py:56-69`` layout: ``<output_dir>/checkpoint-<n>/{config.json, pytorch_model.bin, tokenizer files,
    training_args.bin}`` (+ rotation by ``--save_total_limit``, :41-53).  The optimizer / schedule state goes to
    ``r4d_optimizer.pt`` / ``r4d_scheduler.pt`` in this build's own layout -- NOT under the reference's ``optimizer.pt`` /
    ``scheduler.pt`` names, which ``get_optimizer_scheduler`` (utils/model.py:96-102) would try to ``load_state_dict`` as a
    ``transformers.AdamW`` / ``LambdaLR`` state when pointed at this directory."""
    import glob
    import re
    import shutil
    out = os.path.join(args.output_dir, f"checkpoint-{global_step}")
    os.makedirs(out, exist_ok=True)
    model.save_pretrained(out)
    tokenizer.save_pretrained(out)
    keep = {k: v for k, v in vars(args).items() if isinstance(v, (int, float, str, bool, list, tuple, type(None)))}
    torch.save(keep, os.path.join(out, "training_args.bin"))
    limit = getattr(args, "save_total_limit", None)
    if limit and limit > 0:
        found = []
        for path in glob.glob(os.path.join(args.output_dir, "checkpoint-*")):
            m_ = re.match(r".*checkpoint-([0-9]+)", path)
            if m_:
                found.append((int(m_.group(1)), path))
        for _n, path in sorted(found)[:max(0, len(found) - limit)]:
            shutil.rmtree(path)
    torch.save({"format": "rag4dyg_amd.AdamW", "t": optimizer.t, "lr": optimizer.lr,
                "m": {k: v.cpu() for k, v in optimizer.m.items()}, "v": {k: v.cpu() for k, v in optimizer.v.items()}},
               os.path.join(out, "r4d_optimizer.pt"))
    torch.save({"last_epoch": optimizer.t}, os.path.join(out, "r4d_scheduler.pt"))


def get_training_info(n_batches, args):
    """``train/train_retriever.py:100-118``: training continues from ``--model_name_or_path`` when that is an existing directory
    whose name ends in ``-<global_step>`` (``.../checkpoint-<n>``): (global_step, epochs already trained, optimizer steps to skip
    at the head of an epoch).  Anything else starts at zero."""
    global_step = epochs_trained = steps_trained_in_current_epoch = 0
    path = getattr(args, "model_name_or_path", None)
    if path and os.path.exists(path):
        per_epoch = n_batches // max(1, int(getattr(args, "gradient_accumulation_steps", 1)))
        try:
            global_step = int(path.split("-")[-1].split("/")[0])
            epochs_trained = global_step // per_epoch
            steps_trained_in_current_epoch = global_step % per_epoch
            print("  Continuing training from checkpoint, will skip to saved global_step")
            print("  Continuing training from epoch %d" % epochs_trained)
            print("  Continuing training from global step %d" % global_step)
            print("  Will skip the first %d steps in the first epoch" % steps_trained_in_current_epoch)
        except (ValueError, ZeroDivisionError):
            global_step = epochs_trained = steps_trained_in_current_epoch = 0
            print("  Starting fine-tuning.")
    return global_step, epochs_trained, steps_trained_in_current_epoch


def train_epoch(all_query_time, epoch, model, trainer, optimizer, train_dataloader, global_step, args,
                steps_trained_in_current_epoch=0):
    """``train/train_retriever.py:132-227``: one pass over the shuffled triples.  Returns (global_step, summed loss,
    summed contrastive loss, summed augmentation loss).  ``steps_trained_in_current_epoch`` batches at the head of the pass are
    skipped (:165-167; upstream hands the same count to EVERY epoch of a continued run -- it is passed by value -- and so does
    this)."""
    tr_loss = tr_cl = tr_aug = 0.0
    model.train()                                              # :161 -- dropout on (EncoderTrainer reads model.training)
    i = 0                                                      # counts PROCESSED batches only, as upstream: its `continue` (:165-167)
    for batch in train_dataloader:                             # comes before `i += 1` / `step += 1`, so a skipped batch advances
        if steps_trained_in_current_epoch > 0:                 # neither the lr-schedule index nor the accumulation counter
            steps_trained_in_current_epoch -= 1
            continue
        if args.lrdecay == 1:
            adjust_learning_rate(args, optimizer, epoch, args.learning_rate, i, len(train_dataloader))
        r = training_step(args, model, trainer, optimizer, batch, all_query_time, micro_step=i, sync=False)
        i += 1
        tr_loss = tr_loss + r["loss"]; tr_cl = tr_cl + r["cl_loss"]; tr_aug = tr_aug + r["aug_loss"]   # device sums: no wait per step
        if r["stepped"]:                                       # an optimizer update (train_retriever.py:212-221)
            global_step += 1
        if args.max_steps > 0 and global_step > args.max_steps:
            break
    return global_step, float(tr_loss), float(tr_cl), float(tr_aug)


def distributed_setup(args):
    """(world size, rank) of a run started by ``torch.distributed.run``; joins the process group on first use.  One process
    per GPU over RCCL (backend "nccl"); ``R4D_DIST_BACKEND=gloo`` lets several ranks share one card (tests)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.data_parallel_world = 1
    if world <= 1:
        return 1, 0
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=os.environ.get("R4D_DIST_BACKEND", "nccl"))
    args.data_parallel_world = dist.get_world_size()           # get_dataloader: DistributedSampler (dataloader/retriever.py:160)
    args.data_parallel_rank = dist.get_rank()
    return dist.get_world_size(), dist.get_rank()


def train(args, train_dataset, model, tokenizer):
    """Drop-in for ``train/train_retriever.train`` (``:230-354``): AdamW on the time-decayed contrastive + InfoNCE loss, validation
    hit@3 after every epoch, best checkpoint as ``checkpoint-0`` (only once ``epoch > warmup_steps``, as upstream), last as
    ``checkpoint-1``, early stopping after ``--patience`` epochs without improvement, then the test / validation passes on
    the best and the last weights.  Data parallel: launched under ``torch.distributed.run`` (one process per GPU) every rank
    takes its ``DistributedSampler`` share of the triples and the gradients are averaged with one all-reduce per update (RCCL;
    ``R4D_DIST_BACKEND=gloo`` for ranks that share a card); every rank validates (identical numbers, so the early-stopping
    decision needs no broadcast), rank 0 writes the checkpoints.  Dropout (``model.train()``) draws its masks from the
    library's counter-based generator keyed by ``--seed`` instead of torch's RNG stream.  Continuing from a ``checkpoint-<n>``
    directory given as ``--model_name_or_path`` follows ``get_training_info`` (:100-118) and restores the optimizer state this build
    saved there.  Differences: ``--fp16`` (apex) is not built (raises)."""
    from .dataloader import get_dataloader
    from .retriever import test
    if getattr(args, "fp16", False):
        raise NotImplementedError("retriever training: --fp16 (apex mixed precision) is not built; the path is fp32")
    world, rank = distributed_setup(args)
    train_dataloader, args = get_dataloader(train_dataset, tokenizer, args, split="train")
    gas = max(1, int(getattr(args, "gradient_accumulation_steps", 1)))
    if args.max_steps > 0:
        args.num_train_epochs = args.max_steps // max(1, len(train_dataloader) // gas) + 1
    trainer = EncoderTrainer(model, seed=int(getattr(args, "seed", 0)) + 7919 * rank)      # every rank its own masks
    if world > 1:
        import torch.distributed as dist
        for p in trainer.params.values():                       # DistributedDataParallel's construction-time broadcast
            dist.broadcast(p.data, src=0)
    optimizer = AdamW(trainer.params, trainer.grads, lr=args.learning_rate, eps=args.adam_epsilon, weight_decay=args.weight_decay,
                      flat_grads=trainer.flat_grads)
    print("***** Running training *****")
    print("  Num examples = {}".format(len(train_dataset)))
    print("  Num Epochs = {}".format(args.num_train_epochs))
    print("  Instantaneous batch size per GPU = {}".format(args.per_gpu_train_batch_size))
    all_query_time = torch.load(os.path.join("resources/", args.dataset + '_train_query_time.pt'))     # get_train_query_time.py
    all_query_time = torch.as_tensor(all_query_time).to(args.device)
    # continuing from a checkpoint directory (train_retriever.py:276; optimizer / schedule state: utils/model.py:96-102)
    global_step, epochs_trained, steps_to_skip = get_training_info(len(train_dataloader), args)
    opt_state = os.path.join(args.model_name_or_path, "r4d_optimizer.pt") if getattr(args, "model_name_or_path", None) else None
    if opt_state and os.path.isfile(opt_state):
        optimizer.load_state(torch.load(opt_state, map_location="cpu", weights_only=True))
    tr_loss = 0.0
    best_score, best_epoch, best_state, counter = None, 0, None, 0
    snapshot = lambda: {k: v.detach().clone() for k, v in model.state_dict().items()}
    last_state, epoch = None, epochs_trained
    for epoch in range(epochs_trained, int(args.num_train_epochs)):
        print('==> Training Epoch: ', epoch)
        global_step, ep_loss, cl, au = train_epoch(all_query_time, epoch, model, trainer, optimizer, train_dataloader, global_step, args,
                                                   steps_to_skip)
        tr_loss = ep_loss            # the reference resets tr_loss every epoch (train_retriever.py:158): its logged / returned
                                     # "train_loss" is the LAST epoch's summed loss over the cumulative step count (:303, :354)
        if not math.isfinite(ep_loss):   # the reference trains on in silence; here the validation below refuses NaN weights
            warnings.warn(f"rag4dyg_amd: epoch {epoch}: the summed training loss is {ep_loss} (contrastive {cl}, augmentation {au}) -- "
                          f"e.g. the argparse default --lambda_decay -1 makes the time decay exp(+|dt|) overflow; the reference's "
                          f"scripts pass 0.0001", RuntimeWarning)
        val_metrics, val_loss = test(epoch, args, model, tokenizer, evaluate=True)
        score = val_metrics['hit@3']
        print(f"epoch {epoch}: train_loss {tr_loss / max(global_step, 1):.5f} (cl {cl:.4f} aug {au:.4f}) val_loss {float(val_loss):.5f} "
              f"val_hit@3 {score} lr {optimizer.lr:.3e}")
        print('best_score: ', best_score)
        early_stop = False
        if epoch > args.warmup_steps:
            if best_score is None or score > best_score:
                best_score, best_epoch, best_state, counter = score, epoch, snapshot(), 0
                if rank == 0:
                    save_checkpoint(model, optimizer, tokenizer, args, 0)
            else:
                counter += 1
                print('  EarlyStopping counter: {} out of {}'.format(counter, args.patience))
                early_stop = counter >= args.patience
        if early_stop:
            print('  Early Stopping.....')
            break
        if rank == 0:
            save_checkpoint(model, optimizer, tokenizer, args, 1)
        last_state = snapshot()
    if best_state is None:                        # never past the warm-up epochs: the last weights are the best we have
        best_state, best_epoch = snapshot(), epoch
    last_state = last_state or snapshot()
    print("***** Running testing *****")
    model.load_state_dict(best_state)
    test_metrics = test(best_epoch, args, model, tokenizer, evaluate=False, prefix="best")
    print("test_metrics best epoch : ", test_metrics)
    test(best_epoch, args, model, tokenizer, evaluate=True, prefix="best")
    model.load_state_dict(last_state)
    print("test_metrics last epoch : ", test(epoch, args, model, tokenizer, evaluate=False))
    return global_step, tr_loss / max(global_step, 1)
