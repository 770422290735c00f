"""torch-tensor front end of the C ABI: pointer/shape checks, workspace and stream plumbing only.

Every function requires CUDA(HIP) tensors and raises otherwise -- there is no CPU path here.
"""
import torch

from . import _lib
from ._lib import check


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.R4DError(f"{name}: expected a tensor on the GPU (no CPU fallback in rag4dyg_amd)")
    if t.dtype != dtype:
        raise _lib.R4DError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.R4DError(f"{name}: tensor must be contiguous")
    return t.data_ptr()


_WS = {}


def workspace(nbytes, device, tag="default"):
    """Grow-only scratch buffer per (device, tag); the library itself never allocates."""
    key = (str(device), tag)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


def layernorm(x, w, b, eps=1e-5):
    d = x.shape[-1]
    y = torch.empty_like(x)
    rows = x.numel() // d
    check(_lib.load().r4d_layernorm_f32(_dev(x, torch.float32, "x"), _dev(w, torch.float32, "w"),
                                        _dev(b, torch.float32, "b"), rows, d, eps, y.data_ptr(), _stream()), "layernorm")
    return y


def conv1d(x, w, bias, epilogue="none", residual=None, w_t=None):
    """y = epilogue(x @ W[K,N] + bias); epilogue in {'none','gelu','residual'} (Conv1D, modeling_utils.py:1267-1271).
    ``w_t``: optional contiguous transposed copy W^T [N,K] of the (static) weight -> the k-contiguous GEMM kernel."""
    K, N = w.shape
    M = x.numel() // K
    y = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
    epi = {"none": 0, "gelu": 1, "residual": 2}[epilogue]
    rp = _dev(residual, torch.float32, "residual") if residual is not None else None
    bp = _dev(bias, torch.float32, "bias") if bias is not None else None
    tp = _dev(w_t, torch.float32, "w_t") if w_t is not None else None
    check(_lib.load().r4d_conv1d_f32(_dev(x, torch.float32, "x"), _dev(w, torch.float32, "w"), tp, bp, rp, M, K, N, epi,
                                     y.data_ptr(), _stream()), "conv1d")
    return y


_GEMM_MODES = {"f32": 0, "bf16x3": 1, "f16x2": 2}
_GEMM_MODE = None
DEFAULT_GEMM_MODE = "f16x2"


def gemm_mode():
    """Arithmetic of the encoder's Conv1D GEMMs: ``"f16x2"`` -- fp16 matrix cores, two fp16 terms per operand,
    three products per fp32 product; ``"bf16x3"`` -- bf16 matrix cores, three bf16 terms, six products; ``"f32"`` -- the
    exact-f32 MFMA kernels.  All three at fp32 accuracy (error against float64 no larger than the exact-f32 kernel's).
    Default ``DEFAULT_GEMM_MODE``; ``R4D_GEMM_MODE=f32|bf16x3|f16x2`` (or the older ``R4D_GEMM_SPLIT3=0`` for f32) and
    :func:`set_gemm_mode` select another.  Process-wide: the ranks of a sharded job must agree."""
    global _GEMM_MODE
    if _GEMM_MODE is None:
        import os
        m = os.environ.get("R4D_GEMM_MODE")
        if m is None:
            m = "f32" if os.environ.get("R4D_GEMM_SPLIT3", "1") == "0" else DEFAULT_GEMM_MODE
        set_gemm_mode(m)
    return _GEMM_MODE


_LAST_SPLIT_MODE = None


def set_gemm_mode(mode):
    global _GEMM_MODE, _LAST_SPLIT_MODE
    if mode not in _GEMM_MODES:
        raise ValueError(f"gemm mode {mode!r}: one of {sorted(_GEMM_MODES)}")
    _GEMM_MODE = mode
    if mode != "f32":
        _LAST_SPLIT_MODE = mode
    check(_lib.load().r4d_set_gemm_split3(_GEMM_MODES[mode]), "set_gemm_split3")


def gemm_split3_enabled():
    """True unless the exact-f32 MFMA kernels are selected (``gemm_mode() != "f32"``): the bf16x3 planes are in use (also in
    "f16x2" mode, for the training GEMMs and for layers without fp16 planes)."""
    return gemm_mode() != "f32"


def set_gemm_split3(on):
    """Older on / off switch.  False -> "f32".  True -> the split mode that was selected LAST in this process ("f16x2" or
    "bf16x3": off-then-on returns to where it was), else ``R4D_GEMM_MODE`` if that names a split mode, else
    ``DEFAULT_GEMM_MODE``; a no-op while a split mode is already selected."""
    if not on:
        set_gemm_mode("f32")
    elif gemm_mode() == "f32":
        import os
        m = _LAST_SPLIT_MODE or os.environ.get("R4D_GEMM_MODE", DEFAULT_GEMM_MODE)
        set_gemm_mode(m if m != "f32" else DEFAULT_GEMM_MODE)


H2_MAX_WEIGHT = 6.0e4          # fp16 tops out at 65504: a weight beyond this keeps the bf16x3 planes


def split2_planes(w, transposed=False):
    """Static Conv1D weight [K,N] (``transposed``: an [N,K] copy) -> its two fp16 terms (hi, 2^11-scaled lo) as 128-byte lines,
    a uint16 tensor [N, K/32, 2, 32] (fp16 bit patterns; ``[:, :, 0]`` = hi, ``[:, :, 1]`` = lo of 32 consecutive k): the operand
    format of :func:`conv1d_h2`.  K must be a multiple of 32.  None when the weight exceeds the fp16 range (one host read of the
    tensor's maximum, once per checkpoint)."""
    if transposed:
        N, K = w.shape
    else:
        K, N = w.shape
    amax = float(w.detach().abs().max())
    if not (amax < H2_MAX_WEIGHT):
        return None
    if K % 32:
        raise ValueError(f"split2_planes: K = {K} is not a multiple of 32")
    planes = torch.empty(N, K // 32, 2, 32, dtype=torch.int16, device=w.device)
    check(_lib.load().r4d_split2_planes_f16(_dev(w, torch.float32, "w"), K, N, int(bool(transposed)), planes.data_ptr(),
                                            _stream()), "split2_planes")
    return planes


def set_gemm_h2p(on):
    """f16x2 mode: True (default) -- LayerNorm / GELU write their rows as f16x2 lines and the GEMMs stage both operands by LDS-DMA
    (``csrc/gemm_h2p.hip``); False -- the register-staged ``gemm_h2`` kernels on fp32 activations.  Same bits either way.
    Returns the previous setting."""
    return bool(_lib.load().r4d_set_gemm_h2p(1 if on else 0))


def split2_lines(x):
    """fp32 rows [..., K] -> their f16x2 lines, int16 [rows, K/32, 2, 32] (fp16 bit patterns of x/4: ``[:, :, 0]`` hi, ``[:, :, 1]`` lo'):
    the A operand of :func:`conv1d_h2p` (what the LayerNorm kernels and the c_fc epilogue write in f16x2 mode)."""
    K = x.shape[-1]
    rows = x.numel() // K
    out = torch.empty(rows, K // 32, 2, 32, dtype=torch.int16, device=x.device)
    check(_lib.load().r4d_split2_lines_f16(_dev(x, torch.float32, "x"), rows, K, out.data_ptr(), _stream()), "split2_lines")
    return out


def layernorm_lines(x, w, b, eps=1e-5):
    """:func:`layernorm` whose output rows are f16x2 lines [rows, d/32, 2, 32] (d % 256 == 0)."""
    d = x.shape[-1]
    rows = x.numel() // d
    y = torch.empty(rows, d // 32, 2, 32, dtype=torch.int16, device=x.device)
    check(_lib.load().r4d_layernorm_lines_f32(_dev(x, torch.float32, "x"), _dev(w, torch.float32, "w"), _dev(b, torch.float32, "b"),
                                              rows, d, eps, y.data_ptr(), _stream()), "layernorm_lines")
    return y


def conv1d_h2p(x_lines, planes, bias, epilogue="none", residual=None, out_lines=False):
    """:func:`conv1d_h2` with the input given as f16x2 lines (:func:`split2_lines`), both operands staged by LDS-DMA; bit-identical
    results.  ``out_lines`` (gelu only): the result as lines [M, N/32, 2, 32] instead of fp32 [M, N]."""
    N, K = planes.shape[0], planes.shape[1] * 32
    M = x_lines.shape[0]
    epi = {"none": 0, "gelu": 1, "residual": 2}[epilogue]
    y = (torch.empty(M, N // 32, 2, 32, dtype=torch.int16, device=x_lines.device) if out_lines
         else torch.empty(M, N, dtype=torch.float32, device=x_lines.device))
    rp = _dev(residual, torch.float32, "residual") if residual is not None else None
    bp = _dev(bias, torch.float32, "bias") if bias is not None else None
    check(_lib.load().r4d_conv1d_h2p_f32(_dev(x_lines, torch.int16, "x_lines"), _dev(planes, torch.int16, "planes"), bp, rp, M, K, N, epi,
                                         int(bool(out_lines)), y.data_ptr(), _stream()), "conv1d_h2p")
    return y


def conv1d_h2(x, planes, bias, epilogue="none", residual=None):
    """:func:`conv1d` on the fp16 matrix cores at fp32 accuracy (two fp16 terms per operand, three partial products, two fp32
    accumulator sets); ``planes`` from :func:`split2_planes`.  |x| < 2^18."""
    N, K = planes.shape[0], planes.shape[1] * 32
    M = x.numel() // K
    y = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
    epi = {"none": 0, "gelu": 1, "residual": 2}[epilogue]
    rp = _dev(residual, torch.float32, "residual") if residual is not None else None
    bp = _dev(bias, torch.float32, "bias") if bias is not None else None
    check(_lib.load().r4d_conv1d_h2_f32(_dev(x, torch.float32, "x"), _dev(planes, torch.int16, "planes"), bp, rp, M, K, N, epi,
                                        y.data_ptr(), _stream()), "conv1d_h2")
    return y


def split3_planes(w, transposed=False):
    """Static Conv1D weight [K,N] (``transposed``: an [N,K] copy) -> its three k-contiguous bf16 planes (hi, mid, lo),
    a uint16 tensor [3,N,K] (torch's bfloat16 bit patterns): the operand format of :func:`conv1d_s3`."""
    if transposed:
        N, K = w.shape
    else:
        K, N = w.shape
    planes = torch.empty(3, N, K, dtype=torch.int16, device=w.device)
    check(_lib.load().r4d_split3_planes_bf16(_dev(w, torch.float32, "w"), K, N, int(bool(transposed)), planes.data_ptr(),
                                             _stream()), "split3_planes")
    return planes


def fold_layernorm(wT, ln_w, ln_b):
    """Decode-only copy of a projection that reads a LayerNorm: ``wT`` [N,K] (k-contiguous weight), LayerNorm gain / shift [K]
    -> (``wTg`` [N,K] = gain * wT, ``lnc`` [2,N] = (sum_k gain W, sum_k shift W)): LN(x) W = rstd (x wTg^T - mean c1) + c2."""
    N, K = wT.shape
    wTg = torch.empty_like(wT)
    lnc = torch.empty(2, N, dtype=torch.float32, device=wT.device)
    check(_lib.load().r4d_fold_layernorm_f32(_dev(wT, torch.float32, "wT"), _dev(ln_w, torch.float32, "ln_w"),
                                             _dev(ln_b, torch.float32, "ln_b"), N, K, wTg.data_ptr(), lnc.data_ptr(), _stream()),
          "fold_layernorm")
    return wTg, lnc


def conv1d_s3(x, planes, bias, epilogue="none", residual=None):
    """:func:`conv1d` on the bf16 matrix cores at fp32 accuracy (three-way bf16 split of both operands, six partial
    products, fp32 accumulation); ``planes`` from :func:`split3_planes`."""
    _, N, K = planes.shape
    M = x.numel() // K
    y = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
    epi = {"none": 0, "gelu": 1, "residual": 2}[epilogue]
    rp = _dev(residual, torch.float32, "residual") if residual is not None else None
    bp = _dev(bias, torch.float32, "bias") if bias is not None else None
    check(_lib.load().r4d_conv1d_s3_f32(_dev(x, torch.float32, "x"), _dev(planes, torch.int16, "planes"), bp, rp, M, K, N, epi,
                                        y.data_ptr(), _stream()), "conv1d_s3")
    return y


def set_attention_fused(mode):
    """-1 / None: auto by head_dim (default); True: fused flash-style kernel; False: three-launch GEMM form."""
    _lib.load().r4d_set_attention_fused(-1 if mode is None or mode == -1 else (2 if mode == 2 else int(bool(mode))))


def set_attention_h2(on):
    """In gemm mode "f16x2": head_dim 128 / 256 attention of the encoder on the fp16 matrix cores (csrc/attention_h2.hip; default
    on).  Off: the exact-f32 attention kernels in every mode.  Returns the previous setting.  Process-wide: ranks must agree."""
    return bool(_lib.load().r4d_set_attention_h2(int(bool(on))))


def set_attention_kblk(on):
    """f16x2 attention: K read from the key-blocked image the c_attn GEMM writes (whole cache lines per load
    instruction; default on) or from row-major K words.  Same bits either way.  Returns the previous setting."""
    return bool(_lib.load().r4d_set_attention_kblk(int(bool(on))))


def pack_kblk_words(qkv_words, n_head):
    """Row-major qkv words [B, T, 3d] -> the key-blocked K image (``include/r4d.h``) as a flat int32 tensor."""
    B, T, d3 = qkv_words.shape
    d = d3 // 3
    rows = B * T
    out = torch.empty((rows + 31) // 32 * 32 * d, dtype=torch.int32, device=qkv_words.device)
    check(_lib.load().r4d_pack_kblk_words(_dev(qkv_words, torch.int32, "qkv_words"), rows, n_head, d, out.data_ptr(), _stream()), "pack_kblk_words")
    return out


def attention_h2_kblk(qkv_words, kblk, n_head):
    """:func:`attention_h2` with K taken from ``kblk`` = :func:`pack_kblk_words`."""
    B, T, d3 = qkv_words.shape
    d = d3 // 3
    a = torch.empty(B, T, d, dtype=torch.float32, device=qkv_words.device)
    check(_lib.load().r4d_attention_h2_kblk_f32(_dev(qkv_words, torch.int32, "qkv_words"), _dev(kblk, torch.int32, "kblk"), B, T, n_head, d,
                                                a.data_ptr(), _stream()), "attention_h2_kblk")
    return a


def pack_h2_words(x):
    """fp32 tensor -> its "h2 words" (int32 tensor of the same shape: fp16 hi | fp16 lo' << 16 of value / 4), the operand format of
    :func:`attention_h2` -- what the f16x2 c_attn GEMM writes inside the encoder."""
    x = x.contiguous()
    w = torch.empty(x.shape, dtype=torch.int32, device=x.device)
    check(_lib.load().r4d_pack_h2_words_f32(_dev(x, torch.float32, "x"), x.numel(), w.data_ptr(), _stream()), "pack_h2_words")
    return w


def attention_h2(qkv_words, n_head):
    """:func:`attention` on the fp16 matrix cores at fp32 accuracy: ``qkv_words`` = :func:`pack_h2_words` of the packed c_attn
    output [B,T,3d]; head_dim 128 / 256."""
    B, T, d3 = qkv_words.shape
    d = d3 // 3
    a = torch.empty(B, T, d, dtype=torch.float32, device=qkv_words.device)
    check(_lib.load().r4d_attention_h2_f32(_dev(qkv_words, torch.int32, "qkv_words"), B, T, n_head, d, a.data_ptr(), _stream()),
          "attention_h2")
    return a


def attention(qkv, n_head):
    """Causal MHA over packed c_attn output [B,T,3d] -> [B,T,d] (modeling_gpt2.py:140-175)."""
    B, T, d3 = qkv.shape
    d = d3 // 3
    lib = _lib.load()
    nbytes = lib.r4d_attention_workspace_bytes(B, n_head, T)
    ws = workspace(nbytes, qkv.device, "attn")
    a = torch.empty(B, T, d, dtype=torch.float32, device=qkv.device)
    check(lib.r4d_attention_f32(_dev(qkv, torch.float32, "qkv"), B, T, n_head, d, a.data_ptr(), ws.data_ptr(),
                                ws.numel(), _stream()), "attention")
    return a


def lm_logits(hidden, wte):
    V, d = wte.shape
    M = hidden.numel() // d
    out = torch.empty(hidden.shape[:-1] + (V,), dtype=torch.float32, device=hidden.device)
    check(_lib.load().r4d_lm_logits_f32(_dev(hidden, torch.float32, "hidden"), _dev(wte, torch.float32, "wte"), M, V, d,
                                        out.data_ptr(), _stream()), "lm_logits")
    return out


# ------------------------------------------------------------------------------------------- range guard (include/r4d.h, ABI v6)
RANGE_NONFINITE_HIDDEN, RANGE_BAD_NORM = 1, 2
_RANGE_FLAG = None


def range_flag(device=None):
    """The process's range-guard word (int32 [1] on the GPU), registered with the library on first use (``r4d_set_range_flag``):
    bit 0 = a non-finite row reached ln_f in an ``encode_*`` call (in f16x2 mode: an activation beyond the fp16 range), bit 1 = a
    row that could not be normalised (NaN / inf / zero norm).  Sticky: the library only ORs bits in."""
    global _RANGE_FLAG
    if _RANGE_FLAG is None:
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        _RANGE_FLAG = torch.zeros(1, dtype=torch.int32, device=dev)
        check(_lib.load().r4d_set_range_flag(_RANGE_FLAG.data_ptr()), "set_range_flag")
    return _RANGE_FLAG


def take_range_flag():
    """Read AND clear the range-guard word (one device synchronisation).  0 = every encode / normalise since the last call stayed
    in range."""
    f = range_flag()
    v = int(f.item())
    if v:
        f.zero_()
    return v


def check_range(what):
    """Raise ``R4DError`` if the range-guard word is set (and clear it): the point where results are about to leave the device."""
    v = take_range_flag()
    if v:
        why = []
        if v & RANGE_NONFINITE_HIDDEN:
            why.append("a non-finite hidden state reached ln_f (an activation, q, k or v beyond the arithmetic's range: "
                       "|x| >= 2^18 in gemm mode 'f16x2'; 'bf16x3' and 'f32' have fp32's exponent range)")
        if v & RANGE_BAD_NORM:
            why.append("an embedding row could not be normalised (NaN, inf or zero norm)")
        raise _lib.R4DError(f"{what}: " + "; ".join(why) + " -- no ranking was produced from it")


def normalize_rows(x):
    n, d = x.shape
    range_flag(x.device)
    out = torch.empty_like(x)
    check(_lib.load().r4d_normalize_rows_f32(_dev(x, torch.float32, "x"), n, d, out.data_ptr(), _stream()),
          "normalize_rows")
    return out


SCORE_TOPK_MAX_QUERIES = 65535      # queries per library call (a grid dimension); more are handed over in chunks


def score_topk(q_hat, pool_hat, k, index_offset=0, want_scores=False):
    """(S+1)/2 cosine scan of one pool shard + canonical top-k.  Returns (vals [Q,k], idx int64 [Q,k], scores|None).
    Any number of queries: beyond ``SCORE_TOPK_MAX_QUERIES`` the call is made in chunks (a score does not depend on how the queries
    are batched into calls, so the result is the one call's)."""
    Q, d = q_hat.shape
    if Q > SCORE_TOPK_MAX_QUERIES:
        parts = [score_topk(q_hat[s:s + SCORE_TOPK_MAX_QUERIES].contiguous(), pool_hat, k, index_offset, want_scores)
                 for s in range(0, Q, SCORE_TOPK_MAX_QUERIES)]
        return (torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts]),
                torch.cat([p[2] for p in parts]) if want_scores else None)
    N = pool_hat.shape[0]
    lib = _lib.load()
    ws = workspace(lib.r4d_score_topk_workspace_bytes(Q, N, k), q_hat.device, "score")
    vals = torch.empty(Q, k, dtype=torch.float32, device=q_hat.device)
    idx = torch.empty(Q, k, dtype=torch.int64, device=q_hat.device)
    scores = torch.empty(Q, N, dtype=torch.float32, device=q_hat.device) if want_scores else None
    check(lib.r4d_score_topk_f32(_dev(q_hat, torch.float32, "q_hat"), _dev(pool_hat, torch.float32, "pool_hat"), Q, N, d,
                                 k, int(index_offset), vals.data_ptr(), idx.data_ptr(),
                                 scores.data_ptr() if want_scores else None, ws.data_ptr(), ws.numel(), _stream()),
          "score_topk")
    return vals, idx, scores


def topk_f32(m, k):
    """Canonical per-row top-k of an f32 matrix -> (vals [rows,k], idx int64 [rows,k])."""
    rows, n = m.shape
    lib = _lib.load()
    ws = workspace(lib.r4d_topk_f32_workspace_bytes(rows, n, k), m.device, "topk32")
    vals = torch.empty(rows, k, dtype=torch.float32, device=m.device)
    idx = torch.empty(rows, k, dtype=torch.int64, device=m.device)
    check(lib.r4d_topk_f32(_dev(m, torch.float32, "m"), rows, n, k, vals.data_ptr(), idx.data_ptr(), ws.data_ptr(),
                           ws.numel(), _stream()), "topk_f32")
    return vals, idx


def merge_topk(vals, idx):
    """[G,Q,k] per-shard candidates -> [Q,k] global top-k (same canonical order)."""
    G, Q, k = vals.shape
    ov = torch.empty(Q, k, dtype=torch.float32, device=vals.device)
    oi = torch.empty(Q, k, dtype=torch.int64, device=vals.device)
    lib = _lib.load()
    ws = workspace(lib.r4d_merge_topk_workspace_bytes(G, Q, k), vals.device, "merge")
    check(lib.r4d_merge_topk_f32(_dev(vals, torch.float32, "vals"), _dev(idx, torch.int64, "idx"), G, Q, k,
                                 ov.data_ptr(), oi.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "merge_topk")
    return ov, oi


def argsort_desc(scores):
    """Stable argsort of -scores per row (== np.argsort(-S, axis=1, kind='stable')), int32 [rows,n]."""
    rows, n = scores.shape
    perm = torch.empty(rows, n, dtype=torch.int32, device=scores.device)
    lib = _lib.load()
    f64 = scores.dtype == torch.float64
    fn = lib.r4d_argsort_desc_f64 if f64 else lib.r4d_argsort_desc_f32
    ptr = _dev(scores, torch.float64 if f64 else torch.float32, "scores")
    step = 65535                                            # grid.y limit of one launch
    for r0 in range(0, rows, step):
        nr = min(step, rows - r0)
        ws = workspace(lib.r4d_argsort_workspace_bytes(nr, n, 8 if f64 else 4), scores.device, "argsort")
        check(fn(ptr + r0 * n * scores.element_size(), nr, n, perm.data_ptr() + r0 * n * 4, ws.data_ptr(), ws.numel(),
                 _stream()), "argsort_desc")
    return perm


def jaccard(a_ptr, a_idx, b_ptr, b_idx, vocab, zero_diag=False, sort_rows=None, dense_split=None):
    """f64 [na,nb] Jaccard matrix of CSR sets (occurrence_matrix, retrieval_data_annotation.py:36-41).
    ``dense_split``: the 32 most frequent tokens leave the lists for a per-set membership word -- same values, 2-3x fewer token
    steps on the input sets (ego + <|timeK|> + neighbours).  ``sort_rows``: the kernel visits the A rows longest set first --
    same values, fewer padded token steps.  Both are prepared ON THE DEVICE by ``r4d_jaccard_prepared_f64`` (three small HIP
    launches; until round 5 a dozen torch index ops that cost more GPU time than the kernel).
    Defaults: both only where they pay (sets averaging more than 4 tokens and >= 5*10^6 / 5*10^7 pairs; the 1.8-token output
    sets are bound by the division and the store: the extra lookups cost more than they save)."""
    na, nb = a_ptr.numel() - 1, b_ptr.numel() - 1
    out = torch.empty(na, nb, dtype=torch.float64, device=a_ptr.device)
    if a_idx.numel() == 0:                     # contract: idx buffers hold at least one element
        a_idx = torch.zeros(1, dtype=torch.int32, device=a_ptr.device)
    if b_idx.numel() == 0:
        b_idx = torch.zeros(1, dtype=torch.int32, device=b_ptr.device)
    long_sets = a_idx.numel() > 4 * na
    if dense_split is None:
        dense_split = long_sets and na * nb >= 5_000_000
    if sort_rows is None:
        sort_rows = long_sets and na * nb >= 50_000_000      # small problems: the ordering costs what it saves
    lib = _lib.load()
    ws = None
    if dense_split or sort_rows:
        ws = workspace(lib.r4d_jaccard_prepared_workspace_bytes(na, a_idx.numel(), nb, b_idx.numel(), int(vocab)), a_ptr.device,
                       "jaccard")
    check(lib.r4d_jaccard_prepared_f64(_dev(a_ptr, torch.int32, "a_ptr"), _dev(a_idx, torch.int32, "a_idx"), na, a_idx.numel(),
                                       _dev(b_ptr, torch.int32, "b_ptr"), _dev(b_idx, torch.int32, "b_idx"), nb, b_idx.numel(),
                                       int(vocab), int(bool(zero_diag)), int(bool(dense_split)), int(bool(sort_rows)),
                                       out.data_ptr(), ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
                                       _stream()), "jaccard")
    return out


def topk_f64(m, k):
    rows, n = m.shape
    lib = _lib.load()
    ws = workspace(lib.r4d_topk_f64_workspace_bytes(rows, n, k), m.device, "topk64")
    vals = torch.empty(rows, k, dtype=torch.float64, device=m.device)
    idx = torch.empty(rows, k, dtype=torch.int32, device=m.device)
    check(lib.r4d_topk_f64(_dev(m, torch.float64, "m"), rows, n, k, vals.data_ptr(), idx.data_ptr(), ws.data_ptr(),
                           ws.numel(), _stream()), "topk_f64")
    return vals, idx
