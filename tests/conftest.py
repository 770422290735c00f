import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: tens of seconds of CPU work")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(got, ref):
    """max |got-ref| / max|ref|  -- the 'relative fp32' measure used by every float parity test."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))


def elementwise_err(got, ref, rtol=1e-4, atol=1e-5):
    """Element-wise tolerance check: max over elements of |got-ref| / (rtol * |ref| + atol * max|ref|); PASS iff < 1.
    north_star's "1e-4 relative fp32" applied to every element, with an absolute floor of 1e-5 of the tensor's largest
    magnitude for the elements near zero (a K-term fp32 dot product cannot be relatively accurate where it cancels) --
    unlike ``rel_err`` (max-norm) it cannot hide a relative error on the small elements behind the large ones."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float((np.abs(got - ref) / (rtol * np.abs(ref) + atol * max(np.abs(ref).max(), 1e-30))).max())


def rank_mismatch_report(ref_scores, ref_idx, got_idx):
    """Ranked top-k lists of the device against the reference's stable top-k.  Returns (fraction of rows whose whole
    list is identical, largest REFERENCE-score gap between the item the device ranked at a position and the item the
    reference ranked there, over all mismatching positions).  A mismatch is legitimate only inside fp32 summation noise."""
    ref_scores = np.asarray(ref_scores, dtype=np.float64)
    ref_idx, got_idx = np.asarray(ref_idx), np.asarray(got_idx)
    same = (ref_idx == got_idx)
    gap = 0.0
    for r, c in zip(*np.nonzero(~same)):
        gap = max(gap, abs(ref_scores[r, ref_idx[r, c]] - ref_scores[r, got_idx[r, c]]))
    return float(same.all(axis=1).mean()), float(gap)


def assert_tokens_equal_or_tie(got, want, logits_at, what="", rel_gap=2e-6):
    """Greedy-decoded id lists must be equal.  A first difference at step i is accepted ONLY when the oracle's logits for
    that step (``logits_at(prefix ids) -> 1-D array``) separate the two candidates by less than ``rel_gap * max|logit|``
    (fp32 summation order decides the argmax there); the gap is printed.  Returns 1 when equal, 0 for a justified tie."""
    got, want = list(got), list(want)
    if got == want:
        return 1
    i = next((j for j, (a, b) in enumerate(zip(got, want)) if a != b), min(len(got), len(want)))
    assert i < min(len(got), len(want)), f"{what}: one list is a strict prefix of the other: {got} vs {want}"
    lg = np.asarray(logits_at(want[:i]), dtype=np.float64)
    gap = abs(lg[want[i]] - lg[got[i]])
    scale = np.abs(lg).max()
    print(f"[tie] {what}: step {i} device {got[i]} vs oracle {want[i]}: logit gap {gap:.3e} (max|logit| {scale:.3e})")
    assert gap <= rel_gap * scale, f"{what}: step {i} differs at a logit gap of {gap:.3e} (> {rel_gap} * {scale:.3e})"
    return 0


def load_state_dict_checked(model, sd):
    """``load_state_dict(strict=False)`` with the ONLY tolerated gaps spelled out: the causal-mask buffers ``*.attn.bias`` /
    ``*.attn.masked_bias`` (not parameters; ``.attn.bias`` with the leading dot, so ``attn.c_attn.bias`` does not match) and the
    tied ``lm_head.weight``.  Anything else missing or unexpected fails the test (a fixture that had lost
    both ``c_attn.bias`` tensors loaded silently)."""
    missing, unexpected = model.load_state_dict(sd, strict=False)
    bad = [k for k in missing if not (k.endswith(".attn.bias") or k.endswith(".attn.masked_bias") or k == "lm_head.weight")]
    assert not unexpected, f"unexpected keys: {unexpected}"
    assert not bad, f"missing keys: {bad}"
    return model


GEMM_MODES = ["f16x2", "bf16x3", "f32"]


@pytest.fixture(params=GEMM_MODES)
def gemm_mode(request):
    """Every arithmetic of the encoder's Conv1D GEMMs (``ops.gemm_mode``): fp16 matrix cores / two terms / three products,
    bf16 matrix cores / three terms / six products, exact-f32 MFMA.  Restores the process default afterwards."""
    from rag4dyg_amd import ops
    was = ops.gemm_mode()
    ops.set_gemm_mode(request.param)
    yield request.param
    ops.set_gemm_mode(was)


def g13_state_dict(case, g=None):
    """Weights of one G13 case rebuilt exactly as ``oracle/gen_golden.py g13`` built them for the reference model -- the trained
    G10 tensors (head_dim 128 cases, n_head = 1) or the seeded L2 H2 d512 model (head_dim 256), then the committed
    ``stress_transform`` / ``sharpen_attention`` -- and checked against the fixture's per-tensor bit checksums.
    Returns (state dict without the tied ``lm_head.weight``, n_head)."""
    import torch
    from oracle import gpt2_ref
    g = g if g is not None else load_golden("g13_h2_attention_stress")
    if case.startswith(("hd128", "hd64", "hd32")):
        gw = load_golden("g10_trained_small")
        sd = {n[2:]: torch.from_numpy(gw[n]) for n in gw.files if n.startswith("w:")}
        sd["lm_head.weight"] = sd["transformer.wte.weight"]
    elif case.startswith("hd96"):
        sd = gpt2_ref.make_state_dict(2, 768, 1801, seed=int(g["seed_hd96"]), random_affine=True)
    else:
        sd = gpt2_ref.make_state_dict(2, 512, 1801, seed=int(g["seed_hd256"]), random_affine=True)
    if case != "hd128_plain":
        sd = gpt2_ref.stress_transform(sd)
    if case.endswith("peaked"):
        sd = gpt2_ref.sharpen_attention(sd, 6.0 if case.startswith(("hd256", "hd96")) else 4.0)
    assert np.array_equal(gpt2_ref.weight_bit_checksums(sd), g[case + ":weight_checksums"]), \
        f"G13 {case}: the transforms no longer produce the fixture's weights"
    L, H, d, V, n_pos = (int(x) for x in g[case + ":cfg"])
    assert sd["transformer.wte.weight"].shape == (V, d) and gpt2_ref.n_layers_of(sd) == L
    sd = {n: v for n, v in sd.items() if n != "lm_head.weight"}
    return sd, H
