"""GPU tests of the retriever training step against the reference's own step (tests/golden/g8_training_step.npz:
five forwards, CLtime_loss + alpha * info_nce, parameter gradients from the reference's autograd on CPU), of the optimizer
against the restated transformers.AdamW, of each backward kernel against torch autograd in float64, and of the
``main_retriever.py --do_train`` loop end to end."""
import os
import random
import types

import numpy as np
import pytest
import torch

from conftest import elementwise_err, load_golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def test_device_loss_head_matches_reference_values_and_autograd(dev):
    """csrc/losses.hip (r4d_retriever_losses_f32) against the REFERENCE's CLtime_loss / info_nce values (G8: l4 and l32, computed
    by train/train_retriever.py itself), and its gradient on the [5, B, d] embeddings against float64 autograd of the oracle's
    restatement (oracle/train_ref.py, pinned by the same fixture) -- element-wise; last-batch size (B != per_gpu_train_batch_size:
    the reference rebuilds its mask, the kernel needs none); gradient scale; determinism (same bits on a second launch)."""
    from oracle import train_ref
    from rag4dyg_amd import training
    g = load_golden("g8_training_step")
    T = torch.from_numpy
    for tag in ("l4", "l32"):
        a, p, n = (T(x) for x in g[tag + "_emb"])
        ta, tp, tn = (T(x) for x in g[tag + "_time"])
        B, d = a.shape
        gen = torch.Generator().manual_seed(B)
        s1, s2 = torch.randn(B, d, generator=gen), torch.randn(B, d, generator=gen)
        for alpha, scale in ((1.0, 1.0), (0.3, 0.25)):
            args = types.SimpleNamespace(temperature=0.07, lambda_decay=0.05, alpha=alpha, per_gpu_train_batch_size=B + 1)
            emb = torch.stack([a, p, n, s1, s2]).to(dev)
            losses, demb = training.retriever_losses(args, emb, ta, tp, tn, grad_scale=scale)
            assert abs(losses[0].item() - float(g[tag + "_cltime"])) < 1e-5 * max(1.0, abs(float(g[tag + "_cltime"])))
            # info_nce of the fixture is on (a, p) as the two views
            l2, _ = training.retriever_losses(args, torch.stack([a, p, n, a, p]).to(dev), ta, tp, tn, want_grad=False)
            assert abs(l2[1].item() / (alpha * float(g[tag + "_infonce_raw"])) - 1) < 1e-5
            assert abs(l2[1].item() / (alpha * float(g[tag + "_infonce_rebuilt"])) - 1) < 1e-5
            # gradient: float64 autograd of the oracle's loss functions
            leaf = torch.stack([a, p, n, s1, s2]).double().requires_grad_(True)
            cl = train_ref.cltime_loss(0.07, 0.05, leaf[0], leaf[1], leaf[2], ta.double(), tp.double(), tn.double())
            au = alpha * train_ref.info_nce(leaf[3], leaf[4], 0.07, B)
            ((cl + au) * scale).backward()
            assert abs(losses[2].item() / float((cl + au).item()) - 1) < 1e-5
            ratio = elementwise_err(demb.cpu().numpy(), leaf.grad.numpy())
            print(f"{tag} alpha {alpha}: loss head gradient element-wise ratio {ratio:.3f} (pass < 1), max-norm {rel_err(demb.cpu().numpy(), leaf.grad.numpy()):.2e}")
            assert ratio < 1
            losses_b, demb_b = training.retriever_losses(args, emb, ta, tp, tn, grad_scale=scale)
            assert torch.equal(demb, demb_b) and torch.equal(losses, losses_b)


@pytest.mark.parametrize("tag", ["ts_tiny", "ts_cfg2"])
def test_training_step_forward_losses_equal_reference(dev, tag):
    """anchor / positive / negative / two augmented views through ONE fused launch sequence of the HIP encoder, the
    time-decayed contrastive loss and the InfoNCE term: embeddings within 1e-4 (element-wise), the three loss values within
    1e-4 relative of what the reference's own train_epoch iteration computed (ts_cfg2 = UCI_13 retriever shape L4 H2 d512)."""
    from oracle import gpt2_ref
    from rag4dyg_amd import training
    from rag4dyg_amd.gpt2 import GPT2Config, GPT2LMHeadModelRAG
    g = load_golden("g8_training_step")
    L, H, d, V, pad, B, seed = (int(x) for x in g[tag + "_cfg"])
    eta, gamma, alpha, temp, lam = (float(x) for x in g[tag + "_hyper"])
    sd = gpt2_ref.make_state_dict(L, d, V, n_positions=1024, seed=seed, random_affine=True)
    cfg = GPT2Config(vocab_size=V, n_positions=1024, n_ctx=1024, n_embd=d, n_layer=L, n_head=H)
    cfg.eta, cfg.gamma = eta, gamma
    m = GPT2LMHeadModelRAG(cfg)
    m.load_state_dict(sd, strict=False)
    m = m.to(dev).eval()
    args = types.SimpleNamespace(device=dev, temperature=temp, lambda_decay=lam, alpha=alpha, per_gpu_train_batch_size=B)
    T = torch.from_numpy
    idx = T(g[tag + "_idx"])
    batch = (T(g[tag + "_anchor"]), T(g[tag + "_pos"]), T(g[tag + "_neg"]), idx[:, 0:1], idx[:, 1:2], idx[:, 2:3])
    random.seed(seed)
    r = training.training_step_forward(args, m, batch, T(g[tag + "_times"]).to(dev))
    assert np.array_equal(r["aug1"].cpu().numpy(), g[tag + "_aug1"]) and np.array_equal(r["aug2"].cpu().numpy(), g[tag + "_aug2"])
    emb = r["embeddings"].cpu().numpy()
    assert rel_err(emb, g[tag + "_emb"]) < 1e-4 and elementwise_err(emb, g[tag + "_emb"]) < 1
    got = np.array([r["cl_loss"].item(), r["aug_loss"].item(), r["loss"].item()])
    print(f"{tag}: losses {got} vs reference {g[tag + '_losses']}, embeddings max-norm err {rel_err(emb, g[tag + '_emb']):.2e}")
    assert np.abs(got / g[tag + "_losses"] - 1).max() < 1e-4, (got, g[tag + "_losses"])


@pytest.mark.parametrize("tag", ["ts_tiny", "ts_cfg2"])
def test_training_step_gradients_equal_reference_autograd(dev, tag, gemm_mode):
    """(All three arithmetics -- f16x2: forward and data-gradient GEMMs on the fp16 matrix cores since round 5; bf16x3; exact f32.)
    Backward pass on the HIP kernels against the REFERENCE's autograd (G8: one train_epoch iteration, dropout 0): the
    gradient norm of every parameter tensor and three tensors element-wise from the fixture, and EVERY tensor element-wise
    against the oracle's grad-enabled forward (itself pinned by the same fixture)."""
    from oracle import gpt2_ref, train_ref
    from rag4dyg_amd import training
    from rag4dyg_amd.gpt2 import GPT2Config, GPT2LMHeadModelRAG
    g = load_golden("g8_training_step")
    L, H, d, V, pad, B, seed = (int(x) for x in g[tag + "_cfg"])
    eta, gamma, alpha, temp, lam = (float(x) for x in g[tag + "_hyper"])
    sd = gpt2_ref.make_state_dict(L, d, V, n_positions=1024, seed=seed, random_affine=True)
    cfg = GPT2Config(vocab_size=V, n_positions=1024, n_ctx=1024, n_embd=d, n_layer=L, n_head=H)
    cfg.eta, cfg.gamma = eta, gamma
    m = GPT2LMHeadModelRAG(cfg)
    m.load_state_dict(sd, strict=False)
    m = m.to(dev).eval()
    args = types.SimpleNamespace(device=dev, temperature=temp, lambda_decay=lam, alpha=alpha, per_gpu_train_batch_size=B)
    T = torch.from_numpy
    idx = T(g[tag + "_idx"])
    anchor, pos, neg = T(g[tag + "_anchor"]).to(dev), T(g[tag + "_pos"]).to(dev), T(g[tag + "_neg"]).to(dev)
    random.seed(seed)
    aug1, aug2 = training.aug(anchor, eta, gamma, V - 1)
    trainer = training.EncoderTrainer(m)
    emb = trainer.forward([anchor, pos, neg, aug1, aug2])
    assert rel_err(emb.view(5, B, -1).cpu().numpy(), g[tag + "_emb"]) < 1e-4          # training forward == inference forward
    t = T(g[tag + "_times"])
    losses, demb = training.retriever_losses(args, emb.view(5, B, -1), t[idx[:, 0:1]], t[idx[:, 1:2]], t[idx[:, 2:3]])
    cl, au = losses[0], losses[1]
    grads = trainer.backward(demb.view(5 * B, -1))
    names = [str(n) for n in g[tag + "_grad_names"]]
    got = np.array([grads["transformer.wte.weight" if n == "lm_head.weight" else n].double().norm().item() for n in names])
    worst = np.abs(got / g[tag + "_grad_norms"] - 1).max()
    print(f"{tag}: worst relative error of a parameter's gradient norm {worst:.2e}")
    assert worst < 1e-3, (worst, [n for n, a, b in zip(names, got, g[tag + "_grad_norms"]) if abs(a / b - 1) > 1e-3])
    assert rel_err(grads["transformer.ln_f.weight"].cpu().numpy(), g[tag + "_grad_lnf_w"]) < 1e-3
    assert rel_err(grads["transformer.h.0.attn.c_attn.bias"].cpu().numpy(), g[tag + "_grad_cattn_b0"]) < 1e-3
    assert rel_err(grads["transformer.wte.weight"][:8].cpu().numpy(), g[tag + "_grad_wte_rows"]) < 1e-3
    # ts_tiny: EVERY parameter gradient element-wise against the REFERENCE's autograd: |d| <= 1e-4 |ref| +
    # 1e-5 max|ref| per element (conftest.elementwise_err < 1), the bound the forward tests use -- a wrong bias gradient of a
    # small tensor cannot hide behind a norm
    if tag == "ts_tiny":
        ratios = {}
        for n in names:
            ref = g[tag + "_grad_all:" + n]
            got_t = grads["transformer.wte.weight" if n == "lm_head.weight" else n]
            got_n = (got_t[:64] if n.endswith("wpe.weight") else got_t).cpu().numpy()
            ratios[n] = elementwise_err(got_n, ref)
        worst_n = max(ratios, key=ratios.get)
        print(f"{tag}: element-wise gradient ratio vs the reference autograd, worst {ratios[worst_n]:.3f} ({worst_n}); pass < 1")
        assert ratios[worst_n] < 1, {n: r for n, r in ratios.items() if r >= 1}
    # every tensor element-wise against the oracle's autograd
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "lm_head.weight"}
    sdg["lm_head.weight"] = sdg["transformer.wte.weight"]
    r = train_ref.training_step(sdg, H, anchor.cpu(), pos.cpu(), neg.cpu(), g[tag + "_times"], idx, eta, gamma, alpha, temp, lam,
                                V - 1, seed, with_grad=True)
    r["loss"].backward()
    errs = {n: rel_err(grads[n].cpu().numpy(), sdg[n].grad.numpy()) for n in grads}
    ew = {n: elementwise_err(grads[n].cpu().numpy(), sdg[n].grad.numpy(), rtol=1e-3, atol=1e-4) for n in grads}
    bad = {n: e for n, e in errs.items() if e > 1e-4}
    print(f"{tag}: worst max-norm gradient error {max(errs.values()):.2e} (bound 1e-4); worst element-wise ratio at "
          f"(1e-3 |ref| + 1e-4 max|ref|) {max(ew.values()):.3f} (pass < 1)")
    assert not bad, bad
    assert max(ew.values()) < 1, {n: e for n, e in ew.items() if e >= 1}


def test_training_step_parameter_update_equals_oracle_adamw(dev):
    """One whole training_step (five forwards, losses, backward, clip_grad_norm_, AdamW with decoupled decay) on the device
    against the oracle: reference-pinned gradients (G8) pushed through a CPU restatement of transformers.AdamW."""
    from oracle import gpt2_ref, train_ref
    from rag4dyg_amd import training
    from rag4dyg_amd.gpt2 import GPT2Config, GPT2LMHeadModelRAG
    g = load_golden("g8_training_step")
    tag = "ts_tiny"
    L, H, d, V, pad, B, seed = (int(x) for x in g[tag + "_cfg"])
    eta, gamma, alpha, temp, lam = (float(x) for x in g[tag + "_hyper"])
    sd = gpt2_ref.make_state_dict(L, d, V, n_positions=1024, seed=seed, random_affine=True)
    cfg = GPT2Config(vocab_size=V, n_positions=1024, n_ctx=1024, n_embd=d, n_layer=L, n_head=H)
    cfg.eta, cfg.gamma = eta, gamma
    m = GPT2LMHeadModelRAG(cfg)
    m.load_state_dict(sd, strict=False)
    m = m.to(dev).eval()
    lr, wd, max_norm = 3e-3, 0.01, 1.0
    args = types.SimpleNamespace(device=dev, temperature=temp, lambda_decay=lam, alpha=alpha, per_gpu_train_batch_size=B,
                                 max_grad_norm=max_norm, gradient_accumulation_steps=1)
    T = torch.from_numpy
    idx = T(g[tag + "_idx"])
    batch = (T(g[tag + "_anchor"]), T(g[tag + "_pos"]), T(g[tag + "_neg"]), idx[:, 0:1], idx[:, 1:2], idx[:, 2:3])
    trainer = training.EncoderTrainer(m)
    opt = training.AdamW(trainer.params, trainer.grads, lr=lr, eps=1e-8, weight_decay=wd)
    random.seed(seed)
    r = training.training_step(args, m, trainer, opt, batch, T(g[tag + "_times"]))
    assert abs(r["loss"] / float(g[tag + "_losses"][2]) - 1) < 1e-5
    # oracle AdamW on the DEVICE's gradients (their parity with the reference's autograd is the previous test; at step 1 the
    # update is ~lr * sign(g), which would amplify 1e-5 gradient noise on the near-zero elements into a different test)
    names = list(trainer.params)
    dev_grads = {n: trainer.grads[n].detach().cpu() for n in names}
    coef, total = train_ref.clip_coefficient([dev_grads[n] for n in names], max_norm)
    assert coef < 1.0 and abs(opt.grad_norm() / total - 1) < 1e-5          # the clip is active in this case
    worst = 0.0
    for n in names:
        decay = 0.0 if "bias" in n else wd                                   # "LayerNorm.weight" matches no GPT-2 name: gains decay
        p_ref, m_ref, v_ref = train_ref.adamw_step(sd[n].double(), dev_grads[n].double() * coef, torch.zeros_like(sd[n]).double(),
                                                   torch.zeros_like(sd[n]).double(), 1, lr, (0.9, 0.999), 1e-8, decay)
        step = (p_ref - sd[n].double()).abs().max().item()
        err = (trainer.params[n].detach().cpu().double() - p_ref).abs().max().item()
        worst = max(worst, err / max(step, 1e-12))
        assert rel_err(opt.m[n].cpu().numpy(), m_ref.numpy()) < 1e-5 and rel_err(opt.v[n].cpu().numpy(), v_ref.numpy()) < 1e-5
    print(f"AdamW step: worst |p - p_oracle| relative to the size of the step {worst:.2e}")
    assert worst < 1e-4
    assert "_wt_cache" not in m.transformer.__dict__                         # stale transposed copies dropped


def test_optimizer_state_survives_a_checkpoint_round_trip(dev, tmp_path):
    """``save_checkpoint`` writes the AdamW moments / step / learning rate (``r4d_optimizer.pt``); ``AdamW.load_state`` -- what a
    continued run does with them (utils/model.py:96-102) -- restores them exactly, and refuses another model's file."""
    import types
    from rag4dyg_amd.gpt2 import GPT2Config, GPT2LMHeadModelRAG
    from rag4dyg_amd.training import AdamW, EncoderTrainer, save_checkpoint
    torch.manual_seed(3)
    m = GPT2LMHeadModelRAG(GPT2Config(vocab_size=50, n_positions=32, n_ctx=32, n_embd=64, n_layer=1, n_head=2)).to(dev)
    tr = EncoderTrainer(m, seed=1)
    opt = AdamW(tr.params, tr.grads, lr=1e-3, weight_decay=0.01, flat_grads=tr.flat_grads)
    for g in tr.grads.values():
        g.copy_(torch.randn_like(g) * 0.01)
    opt.step(max_grad_norm=1.0); opt.step(max_grad_norm=1.0)
    opt.lr = 7e-4

    class Tok:
        def save_pretrained(self, d):
            pass
    args = types.SimpleNamespace(output_dir=str(tmp_path), save_total_limit=None)
    save_checkpoint(m, opt, Tok(), args, 5)
    state = torch.load(os.path.join(str(tmp_path), "checkpoint-5", "r4d_optimizer.pt"), map_location="cpu", weights_only=True)
    opt2 = AdamW(tr.params, tr.grads, lr=1e-3, weight_decay=0.01, flat_grads=tr.flat_grads)
    opt2.load_state(state)
    assert opt2.t == 2 and opt2.lr == 7e-4
    assert all(torch.equal(opt2.m[n], opt.m[n]) and torch.equal(opt2.v[n], opt.v[n]) for n in opt.m)
    bad = dict(state, m={k + "_x": v for k, v in state["m"].items()})
    with pytest.raises(Exception):
        opt2.load_state(bad)


def test_non_finite_gradient_stays_loud_in_the_deterministic_embedding_sum(dev):
    """the token-embedding gradient is a 64-bit fixed-point sum (bit-reproducible); a NaN / Inf contribution has no
    fixed-point image and used to become 0 or a saturated integer -- a finite, WRONG gradient.  Now it poisons the table: the whole
    wte gradient (and with it the gradient norm the clip uses) is NaN, as with the float atomics of a framework backward."""
    from oracle import gpt2_ref
    from rag4dyg_amd import training
    from rag4dyg_amd.gpt2 import GPT2Config, GPT2LMHeadModelRAG
    L, H, d, V = 1, 2, 64, 50
    cfg = GPT2Config(vocab_size=V, n_positions=64, n_ctx=64, n_embd=d, n_layer=L, n_head=H)
    m = GPT2LMHeadModelRAG(cfg)
    m.load_state_dict(gpt2_ref.make_state_dict(L, d, V, n_positions=64, seed=1, random_affine=True), strict=False)
    m = m.to(dev).eval()
    ids = torch.randint(0, V, (4, 9), generator=torch.Generator().manual_seed(0)).to(dev)
    trainer = training.EncoderTrainer(m)
    emb = trainer.forward([ids])
    good = trainer.backward(torch.ones_like(emb))["transformer.wte.weight"].clone()
    assert torch.isfinite(good).all() and good.abs().max() > 0
    trainer.forward([ids])
    bad_in = torch.ones_like(emb)
    bad_in[2, 5] = float("inf")
    bad = trainer.backward(bad_in)["transformer.wte.weight"]
    assert torch.isnan(bad).all()
    opt = training.AdamW(trainer.params, trainer.grads, lr=1e-3, flat_grads=trainer.flat_grads)
    opt.sumsq.zero_()
    assert not np.isfinite(float(trainer.flat_grads.double().pow(2).sum().sqrt()))
    # NO run of same-sign contributions may wrap the 64-bit sum into a finite wrong value (round 4's fixed 2^17 bound
    # with a 2^44 scale let 36 terms of ~2^61 wrap silently).  The fixed-point scale now follows max |contribution| of the step
    # and the number of rows, so every magnitude has an image: one token in every position, upstream gradient 1e5 / 1e-20 / 1e30
    # each -- the sums are finite, linear in the upstream gradient (the scale is a power of two) and the same bits on a second run
    same = torch.full((4, 9), 7, dtype=torch.int64, device=dev)
    trainer.forward([same])
    unit = trainer.backward(torch.ones_like(emb))["transformer.wte.weight"].clone()
    assert torch.isfinite(unit).all() and unit[7].abs().max() > 0 and float(unit[:7].abs().max()) == 0.0 == float(unit[8:].abs().max())
    for scale in (1e5, 1e-20, 1e30):
        trainer.forward([same])
        big = trainer.backward(torch.full_like(emb, scale))["transformer.wte.weight"].clone()
        assert torch.isfinite(big).all(), scale
        rel = ((big[7].double() - scale * unit[7].double()).abs().max() / (scale * unit[7].double().abs().max())).item()
        assert rel < 1e-5, (scale, rel)                                  # the encoder backward is linear in its upstream gradient
        trainer.forward([same])
        again = trainer.backward(torch.full_like(emb, scale))["transformer.wte.weight"]
        assert torch.equal(again, big), scale


@pytest.mark.parametrize("split_mode", ["f16x2", "bf16x3"])
@pytest.mark.parametrize("via_step", [True, False])
def test_encode_after_optimizer_step_uses_the_updated_weights(dev, via_step, split_mode):
    """``r4d_adamw_step_f32`` writes the parameters through raw pointers, so torch's version counters do not
    move and the inference path's derived weights (transposed copies, bf16x3 planes -- default ON --, LayerNorm-folded decode
    weights) would stay at the values of the first validation.  Encode, take one optimizer step with a LARGE learning rate
    (through ``training_step`` and through a bare ``AdamW.step``), then the model's encode -- and one cached decode step -- must equal
    the same calls on a fresh copy of the updated weights -- under BOTH split arithmetics (each has its own plane cache)."""
    from oracle import gpt2_ref
    from rag4dyg_amd import ops, training
    from rag4dyg_amd.gpt2 import GPT2Config, GPT2LMHeadModelRAG
    g = load_golden("g8_training_step")
    tag = "ts_tiny"
    _L, _H, _d, V0, _pad, B, seed = (int(x) for x in g[tag + "_cfg"])
    L, H, d, V = 2, 2, 512, V0                                     # d 512: the decode step's LayerNorm-folded projections exist
    eta, gamma, alpha, temp, lam = (float(x) for x in g[tag + "_hyper"])
    cfg = GPT2Config(vocab_size=V, n_positions=128, n_ctx=128, n_embd=d, n_layer=L, n_head=H)
    cfg.eta, cfg.gamma = eta, gamma
    sd = gpt2_ref.make_state_dict(L, d, V, n_positions=128, seed=seed, random_affine=True)
    m = GPT2LMHeadModelRAG(cfg)
    m.load_state_dict(sd, strict=False)
    m = m.to(dev).eval()
    was = ops.gemm_mode()
    ops.set_gemm_mode(split_mode)                                  # BOTH plane caches: _h2_cache (f16x2) and _w3_cache (bf16x3)
    try:
        T = torch.from_numpy
        idx = T(g[tag + "_idx"])
        ids = T(g[tag + "_anchor"]).to(dev)
        kv = m.transformer.new_kv_cache(ids.shape[0], 64, dev)
        pos0 = torch.full((ids.shape[0],), ids.shape[1], dtype=torch.int32, device=dev)
        before = m.transformer.encode(ids, want_hidden=False, want_meanpool=True)["meanpool"].clone()
        m.transformer.prefill(kv, input_ids=ids)
        m.transformer.decode_step(kv, pos0, input_ids=ids[:, -1])             # builds the folded decode weights
        trainer = training.EncoderTrainer(m)
        opt = training.AdamW(trainer.params, trainer.grads, lr=0.05, eps=1e-8, weight_decay=0.0)
        args = types.SimpleNamespace(device=dev, temperature=temp, lambda_decay=lam, alpha=alpha, per_gpu_train_batch_size=B,
                                     max_grad_norm=0.0, gradient_accumulation_steps=1)
        batch = (T(g[tag + "_anchor"]), T(g[tag + "_pos"]), T(g[tag + "_neg"]), idx[:, 0:1], idx[:, 1:2], idx[:, 2:3])
        random.seed(seed)
        if via_step:
            training.training_step(args, m, trainer, opt, batch, T(g[tag + "_times"]))
        else:
            emb = trainer.forward([ids])
            trainer.backward(torch.ones_like(emb))
            opt.step(0.0)
        after = m.transformer.encode(ids, want_hidden=False, want_meanpool=True)["meanpool"]
        h_last = m.transformer.prefill(kv, input_ids=ids)
        dec = m.transformer.decode_step(kv, pos0, input_ids=ids[:, -1])
        fresh = GPT2LMHeadModelRAG(cfg)
        fresh.load_state_dict({k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, strict=False)
        fresh = fresh.to(dev).eval()
        want = fresh.transformer.encode(ids, want_hidden=False, want_meanpool=True)["meanpool"]
        kv2 = fresh.transformer.new_kv_cache(ids.shape[0], 64, dev)
        fresh.transformer.prefill(kv2, input_ids=ids)
        want_dec = fresh.transformer.decode_step(kv2, pos0, input_ids=ids[:, -1])
    finally:
        ops.set_gemm_mode(was)
    moved = (after - before).abs().max().item() / before.abs().max().item()
    assert moved > 1e-2, f"the step did not move the embeddings ({moved:.2e}): the test would prove nothing"
    assert torch.equal(after, want), f"stale derived weights after the optimizer step: {(after - want).abs().max().item():.3e}"
    assert torch.equal(dec, want_dec), f"stale folded decode weights: {(dec - want_dec).abs().max().item():.3e}"


@pytest.mark.parametrize("L,H,d,V,B,Ts", [(2, 6, 768, 300, 4, (37, 50, 23)),      # wikiv2 script shape (head_dim 128)
                                           (3, 2, 256, 200, 5, (21, 33, 40)),      # hepth script shape (head_dim 128, d = 256)
                                           (1, 8, 512, 500, 3, (130, 9, 64)),      # reddit shape (head_dim 64), one batch > 128 positions
                                           (2, 4, 64, 60, 2, (25, 30, 22))])       # head_dim 16
def test_training_gradients_other_shapes_equal_oracle(dev, L, H, d, V, B, Ts, gemm_mode):
    """(All three arithmetics.)  The training forward + backward at the model shapes of the other reference scripts (wikiv2, hepth, reddit; ragged padded
    lengths, also above one 128-position attention tile) against the oracle's grad-enabled forward + CPU autograd (pinned by G8):
    embeddings, the loss and every parameter gradient element-wise (max-norm)."""
    from oracle import gpt2_ref, train_ref
    from rag4dyg_amd import training
    from rag4dyg_amd.gpt2 import GPT2Config, GPT2LMHeadModelRAG
    seed = L * 100 + d
    sd = gpt2_ref.make_state_dict(L, d, V, n_positions=1024, seed=seed, random_affine=True)
    cfg = GPT2Config(vocab_size=V, n_positions=1024, n_ctx=1024, n_embd=d, n_layer=L, n_head=H)
    cfg.eta, cfg.gamma = 0.8, 0.4
    m = GPT2LMHeadModelRAG(cfg)
    m.load_state_dict(sd, strict=False)
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(seed)
    pad = V - 2

    def batch(T):
        ids = torch.randint(0, V - 2, (B, T), generator=g)
        for i in range(B):                                        # right-padded ragged rows
            n = int(torch.randint(max(5, T // 2), T + 1, (1,), generator=g))
            ids[i, n:] = pad
        ids[0, :] = torch.randint(0, V - 2, (T,), generator=g)      # one full-length row
        return ids
    anchor, pos, neg = batch(Ts[0]), batch(Ts[1]), batch(Ts[2])
    times = torch.rand(3 * B, generator=g) * 50
    idx = torch.arange(3 * B).view(3, B).t().contiguous()
    alpha, temp, lam = 0.7, 0.2, 0.05
    args = types.SimpleNamespace(device=dev, temperature=temp, lambda_decay=lam, alpha=alpha, per_gpu_train_batch_size=B)
    random.seed(seed)
    aug1, aug2 = training.aug(anchor, cfg.eta, cfg.gamma, V - 1)
    trainer = training.EncoderTrainer(m)
    emb = trainer.forward([t.to(dev) for t in (anchor, pos, neg, aug1, aug2)])
    losses, demb = training.retriever_losses(args, emb.view(5, B, -1), times[idx[:, 0:1]], times[idx[:, 1:2]], times[idx[:, 2:3]])
    cl, au = losses[0], losses[1]
    grads = trainer.backward(demb.view(5 * B, -1))
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "lm_head.weight"}
    sdg["lm_head.weight"] = sdg["transformer.wte.weight"]
    r = train_ref.training_step(sdg, H, anchor, pos, neg, times, idx, cfg.eta, cfg.gamma, alpha, temp, lam, V - 1, seed, with_grad=True)
    assert rel_err(emb.view(5, B, -1).cpu().numpy(), r["emb"].detach().numpy()) < 1e-4
    assert abs(float((cl + au).item()) / float(r["loss"].item()) - 1) < 1e-4
    r["loss"].backward()
    errs = {n: rel_err(grads[n].cpu().numpy(), sdg[n].grad.numpy()) for n in grads}
    print(f"L{L} H{H} d{d}: worst element-wise (max-norm) gradient error {max(errs.values()):.2e}")
    assert max(errs.values()) < 1e-3, {n: e for n, e in errs.items() if e > 1e-3}
    # run-to-run determinism, bit for bit: every sum of the step has a fixed order (the token-embedding gradient, whose rows
    # collide by data, is accumulated in 64-bit fixed point; the reference's GPU embedding backward is not reproducible)
    first = {n: t.clone() for n, t in grads.items()}
    for _ in range(2):
        emb2 = trainer.forward([t.to(dev) for t in (anchor, pos, neg, aug1, aug2)])
        grads2 = trainer.backward(demb.view(5 * B, -1))
        assert torch.equal(emb2, emb)
        assert all(torch.equal(grads2[n], first[n]) for n in first), [n for n in first if not torch.equal(grads2[n], first[n])]


def test_three_training_steps_track_the_oracle(dev):
    """Three consecutive updates (clip + AdamW, lr large enough to move the weights visibly) on the device against the oracle's
    loop (grad-enabled forward, CPU autograd, restated clip_grad_norm_ + transformers.AdamW): the loss of EVERY step -- steps 2
    and 3 see the updated weights, so a stale transposed copy or optimizer state would show -- and the final parameters."""
    from oracle import gpt2_ref, train_ref
    from rag4dyg_amd import training
    m, args, batch, times, mask, seed = _tiny_step_inputs(dev)
    _, _, batch2, _, _, _ = _tiny_step_inputs(dev, roll=1)
    g = load_golden("g8_training_step")
    tag = "ts_tiny"
    L, H, d, V, pad, B, _ = (int(x) for x in g[tag + "_cfg"])
    eta, gamma, alpha, temp, lam = (float(x) for x in g[tag + "_hyper"])
    lr, wd, max_norm = 2e-3, 0.01, 1.0
    trainer = training.EncoderTrainer(m)
    opt = training.AdamW(trainer.params, trainer.grads, lr=lr, eps=1e-8, weight_decay=wd, flat_grads=trainer.flat_grads)
    sd = gpt2_ref.make_state_dict(L, d, V, n_positions=1024, seed=seed, random_affine=True)
    P = {k: v.clone().double() for k, v in sd.items() if k != "lm_head.weight"}
    M_ = {k: torch.zeros_like(v) for k, v in P.items()}
    V_ = {k: torch.zeros_like(v) for k, v in P.items()}
    for step, b in enumerate((batch, batch2, batch), start=1):
        random.seed(seed + step)
        r = training.training_step(args, m, trainer, opt, b, times)
        sdg = {k: v.float().requires_grad_(True) for k, v in P.items()}
        sdg["lm_head.weight"] = sdg["transformer.wte.weight"]
        idx = torch.cat(b[3:6], dim=1)
        ref = train_ref.training_step(sdg, H, b[0], b[1], b[2], times, idx, eta, gamma, alpha, temp, lam, V - 1, seed + step,
                                      with_grad=True)
        assert abs(r["loss"] / float(ref["loss"].item()) - 1) < 2e-5, (step, r["loss"], float(ref["loss"].item()))
        ref["loss"].backward()
        names = [k for k in P if sdg[k].grad is not None]
        coef, _ = train_ref.clip_coefficient([sdg[k].grad for k in names], max_norm)
        for k in names:
            decay = 0.0 if "bias" in k else wd
            P[k], M_[k], V_[k] = train_ref.adamw_step(P[k], sdg[k].grad.double() * coef, M_[k], V_[k], step, lr, (0.9, 0.999), 1e-8, decay)
    worst = max(rel_err(trainer.params[k].detach().cpu().numpy(), P[k].numpy()) for k in trainer.params)
    print(f"three steps: worst parameter difference {worst:.2e} (relative to the tensor's largest magnitude)")
    # Adam normalises every element's step to ~lr whatever the size of its gradient, so the 3e-5 gradient noise between the two
    # implementations decides part of the step of the near-zero-gradient elements: a few 1e-4 after three steps of lr = 2e-3
    # (measured 2.1e-4); a stale weight copy or optimizer state is a loss mismatch above and >= 1e-2 here
    assert worst < 1e-3


def test_main_retriever_do_train_end_to_end(dev, tmp_path, monkeypatch):
    """``main_retriever.py --do_train`` on a toy dataset in the reference's file grammar: annotation triples from this
    build's retrieval_data_annotation CLI, query times, a few epochs of the training loop (validation after every epoch,
    best / last checkpoints in the reference layout), then ``--do_eval`` on what it wrote.  The loss falls."""
    import io
    import re
    from contextlib import redirect_stdout
    import main_retriever
    from test_gpu_pipeline import _write_dataset
    from rag4dyg_amd import annotation
    base = _write_dataset(str(tmp_path), n_train=96, n_val=40, n_test=37, seed=4)
    monkeypatch.chdir(tmp_path)
    np.random.seed(1)
    annotation.main(["retrieval_data_annotation.py", "toy", "4", "0.3"])
    ret = tmp_path / "resources" / "toy" / "4" / "train_retrieval"
    assert (ret / "train_index.retrieval").stat().st_size > 0
    torch.save(torch.rand(96) * 20, tmp_path / "resources" / "toy_train_query_time.pt")
    out = tmp_path / "out"
    out.mkdir()
    common = (f"--dataset toy --timestamp 4 --output_dir {out} --model_type gpt2 --model_name_or_path gpt2 "
              f"--train_data_file {base}/train.link_prediction --train_pair_data_file {ret}/train_index.retrieval "
              f"--eval_data_file {base}/val.link_prediction --eval_data_gt_file {ret}/val_score.retrieval "
              f"--test_data_file {base}/test.link_prediction --test_data_gt_file {ret}/test_score.retrieval "
              f"--block_size 512 --n_layer 2 --n_head 2 --n_embed 64 --topK 5 --seed 3")
    argv = (common + " --do_train --per_gpu_train_batch_size 16 --num_train_epochs 4 --learning_rate 2e-3 --warmup_steps 0 "
                     "--lambda_decay 0.05 --alpha 0.1 --temperature 0.2 --patience 10").split()
    buf = io.StringIO()
    with redirect_stdout(buf):
        main_retriever.main(argv)
    log = buf.getvalue()
    losses = [float(x) for x in re.findall(r"epoch \d+: train_loss ([0-9.]+)", log)]
    assert len(losses) == 4 and all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    for ck in ("checkpoint-0", "checkpoint-1"):
        for f in ("config.json", "pytorch_model.bin", "tokenizer.json", "training_args.bin", "r4d_optimizer.pt", "r4d_scheduler.pt"):
            assert (out / ck / f).exists(), (ck, f)
    sd = torch.load(out / "checkpoint-1" / "pytorch_model.bin", map_location="cpu", weights_only=True)
    assert "transformer.h.1.mlp.c_proj.weight" in sd and "lm_head.weight" in sd and all(torch.isfinite(v).all() for v in sd.values())
    assert "test_metrics best epoch" in log and "test_metrics last epoch" in log
    res = tmp_path / "resources" / "retrieval_result" / "toy"
    assert (res / "test_index.gen").exists() and (res / "val_index.gen").exists()
    buf = io.StringIO()
    with redirect_stdout(buf):
        main_retriever.main((common + " --do_eval --eval_all_checkpoints").split())
    assert buf.getvalue().count("test_metrics:") == 2


def _tiny_step_inputs(dev, roll=0):
    from oracle import gpt2_ref
    from rag4dyg_amd import training
    from rag4dyg_amd.gpt2 import GPT2Config, GPT2LMHeadModelRAG
    g = load_golden("g8_training_step")
    tag = "ts_tiny"
    L, H, d, V, pad, B, seed = (int(x) for x in g[tag + "_cfg"])
    eta, gamma, alpha, temp, lam = (float(x) for x in g[tag + "_hyper"])
    sd = gpt2_ref.make_state_dict(L, d, V, n_positions=1024, seed=seed, random_affine=True)
    cfg = GPT2Config(vocab_size=V, n_positions=1024, n_ctx=1024, n_embd=d, n_layer=L, n_head=H)
    cfg.eta, cfg.gamma = eta, gamma
    m = GPT2LMHeadModelRAG(cfg)
    m.load_state_dict(sd, strict=False)
    m = m.to(dev).eval()
    T = torch.from_numpy
    idx = T(g[tag + "_idx"])
    # roll != 0: a different batch (the anchors, and their times, shifted against the positives / negatives)
    batch = (torch.roll(T(g[tag + "_anchor"]), roll, 0), T(g[tag + "_pos"]), T(g[tag + "_neg"]), torch.roll(idx[:, 0:1], roll, 0),
             idx[:, 1:2], idx[:, 2:3])
    args = types.SimpleNamespace(device=dev, temperature=temp, lambda_decay=lam, alpha=alpha, per_gpu_train_batch_size=B,
                                 max_grad_norm=1.0, gradient_accumulation_steps=1)
    return m, args, batch, T(g[tag + "_times"]), None, seed


def test_gradient_accumulation_equals_mean_of_micro_batch_gradients(dev):
    """--gradient_accumulation_steps 2 (train_retriever.py:202-212): no update after the first micro-step, and the gradient the
    optimizer sees after the second is (g1 + g2) / 2 of the two micro-batches' own gradients."""
    from rag4dyg_amd import training
    m, args, b1, times, mask, seed = _tiny_step_inputs(dev)
    _, _, b2, _, _, _ = _tiny_step_inputs(dev, roll=1)
    singles = []
    for b in (b1, b2):                                           # each micro-batch alone (lr 0: the parameters stay put)
        tr = training.EncoderTrainer(m)
        opt = training.AdamW(tr.params, tr.grads, lr=0.0, flat_grads=tr.flat_grads)
        random.seed(seed)
        training.training_step(args, m, tr, opt, b, times)
        singles.append(tr.flat_grads.clone())
    args.gradient_accumulation_steps = 2
    tr = training.EncoderTrainer(m)
    opt = training.AdamW(tr.params, tr.grads, lr=0.0, flat_grads=tr.flat_grads)
    random.seed(seed)
    r1 = training.training_step(args, m, tr, opt, b1, times, micro_step=0)
    assert not r1["stepped"] and opt.t == 0
    random.seed(seed)
    r2 = training.training_step(args, m, tr, opt, b2, times, micro_step=1)
    assert r2["stepped"] and opt.t == 1
    want = (singles[0].double() + singles[1].double()) / 2
    assert rel_err(tr.flat_grads.cpu().numpy(), want.cpu().numpy()) < 1e-6
    assert float(tr.flat_accum.abs().max()) == 0.0               # model.zero_grad()


_DP_WORKER = r'''
import os, random, sys, types
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from test_gpu_training import _tiny_step_inputs
from rag4dyg_amd import training
rank = int(os.environ["RANK"])
dist.init_process_group(backend="gloo")
dev = torch.device("cuda:0")
m, args, batch, times, mask, seed = _tiny_step_inputs(dev, roll=rank)          # every rank its own batch
tr = training.EncoderTrainer(m)
opt = training.AdamW(tr.params, tr.grads, lr=1e-3, weight_decay=0.01, flat_grads=tr.flat_grads)
random.seed(seed)
# the local gradient is copied right before the collective
orig = tr.all_reduce_mean
local = {}
def spy():
    local["g"] = tr.flat_grads.clone()
    orig()
tr.all_reduce_mean = spy
training.training_step(args, m, tr, opt, batch, times)
both = [torch.empty_like(local["g"]) for _ in range(2)]
dist.all_gather(both, local["g"])
want = (both[0].double() + both[1].double()) / 2
err = float((tr.flat_grads.double() - want).abs().max() / want.abs().max())
assert float((both[0] - both[1]).abs().max()) > 0, "the two ranks saw the same batch"
assert err < 1e-6, err
flat_p = torch.cat([p.detach().reshape(-1) for p in tr.params.values()])
ps = [torch.empty_like(flat_p) for _ in range(2)]
dist.all_gather(ps, flat_p)
assert torch.equal(ps[0], ps[1]), "parameters diverged across ranks"
print("DP_OK", err)
dist.destroy_process_group()
'''


def test_data_parallel_step_averages_gradients_over_ranks(dev, tmp_path):
    """Two ranks (gloo here: they share the one GPU; RCCL on a node) run one training step on DIFFERENT batches: the gradient
    each optimizer sees is the mean of the two local gradients (one all-reduce over the flat buffer) and the updated
    parameters are bit-identical on both ranks (DistributedDataParallel's contract, train_retriever.py:260-266)."""
    import os
    import subprocess
    import sys
    from conftest import REPO
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER)
    procs = [subprocess.Popen([sys.executable, str(script), REPO],
                              env=dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                                       MASTER_PORT="29541"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs) and all("DP_OK" in o for o in outs), [o[-2000:] for o in outs]


def test_main_retriever_do_train_two_ranks(dev, tmp_path):
    """``main_retriever.py --do_train`` as torch.distributed.run starts it (one process per rank; gloo on this one-GPU box):
    DistributedSampler shares of the triples, averaged gradients, every rank validates, rank 0 writes the checkpoints."""
    import os
    import subprocess
    import sys
    from conftest import REPO
    from test_gpu_pipeline import _write_dataset
    from rag4dyg_amd import annotation
    base = _write_dataset(str(tmp_path), n_train=96, n_val=40, n_test=37, seed=4)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        np.random.seed(1)
        annotation.main(["retrieval_data_annotation.py", "toy", "4", "0.3"])
    finally:
        os.chdir(cwd)
    ret = tmp_path / "resources" / "toy" / "4" / "train_retrieval"
    torch.save(torch.rand(96) * 20, tmp_path / "resources" / "toy_train_query_time.pt")
    out = tmp_path / "out"
    out.mkdir()
    argv = (f"--dataset toy --timestamp 4 --output_dir {out} --model_type gpt2 --model_name_or_path gpt2 "
            f"--train_data_file {base}/train.link_prediction --train_pair_data_file {ret}/train_index.retrieval "
            f"--eval_data_file {base}/val.link_prediction --eval_data_gt_file {ret}/val_score.retrieval "
            f"--test_data_file {base}/test.link_prediction --test_data_gt_file {ret}/test_score.retrieval "
            f"--block_size 512 --n_layer 2 --n_head 2 --n_embed 64 --topK 5 --seed 3 --do_train --per_gpu_train_batch_size 16 "
            f"--num_train_epochs 3 --learning_rate 2e-3 --warmup_steps 0 --lambda_decay 0.05 --alpha 0.1 --temperature 0.2 "
            f"--patience 10 --gradient_accumulation_steps 2").split()
    procs = []
    for rk in range(2):
        env = dict(os.environ, R4D_DIST_BACKEND="gloo", PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""),
                   RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29542")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "main_retriever.py")] + argv, cwd=tmp_path, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    done = [pr.communicate(timeout=900) for pr in procs]
    assert all(pr.returncode == 0 for pr in procs), [e[-2500:] for _, e in done]
    import re
    per_rank = [[float(x) for x in re.findall(r"val_hit@3 ([0-9.eE+-]+)", o)] for o, _ in done]
    assert len(per_rank[0]) == 3 and per_rank[0] == per_rank[1]                # same weights on both ranks -> same validation
    assert (out / "checkpoint-1" / "pytorch_model.bin").exists() and (out / "checkpoint-1" / "r4d_optimizer.pt").exists()
    assert all("test_metrics last epoch" in o for o, _ in done)


def test_real_uci13_training_two_epochs(dev, tmp_path, monkeypatch):
    """The real UCI_13/12 data (G5 / G6 / G9 fixtures: 1,708 training histories, the reference's 9,578 annotation triples and its
    query times) through ``main_retriever.py --do_train`` with the flags of scripts/train_retriever/train_retriever_UCI_13.sh
    (L4 H2 d512, batch 64, lr 1e-5, dropout on), two epochs: 300 updates at the script's shape, losses finite, the InfoNCE term
    falls by more than 3x between the epochs (61 -> ~4 per step when measured), checkpoints written, both final test passes run."""
    import importlib.util
    import io
    import re
    from contextlib import redirect_stdout
    from conftest import REPO
    import main_retriever
    spec = importlib.util.spec_from_file_location("train_uci13_demo", os.path.join(REPO, "tools", "train_uci13_demo.py"))
    demo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo)
    base, ret = demo.build_workdir(str(tmp_path))
    monkeypatch.chdir(tmp_path)
    out = tmp_path / "out"
    argv = (f"--dataset UCI_13 --timestamp 12 --eta 0.8 --gamma 0.4 --temperature 0.1 --alpha 1 --lambda_decay 0.0001 --lrdecay 1 "
            f"--warmup_steps 0 --output_dir {out} --model_type gpt2 --model_name_or_path gpt2 --train_data_file {base}/train.link_prediction "
            f"--train_pair_data_file {ret}/train_index.retrieval --do_train --eval_data_file {base}/val.link_prediction "
            f"--eval_data_gt_file {ret}/val_score.retrieval --test_data_file {base}/test.link_prediction "
            f"--test_data_gt_file {ret}/test_score.retrieval --per_gpu_train_batch_size 64 --num_train_epochs 2 "
            f"--learning_rate 1e-5 --n_layer 4 --n_head 2 --n_embed 512 --block_size 512 --seed 42 --patience 50 --topK 5").split()
    buf = io.StringIO()
    with redirect_stdout(buf):
        main_retriever.main(argv)
    log = buf.getvalue()
    ep = re.findall(r"epoch (\d+): train_loss ([0-9.]+) \(cl ([0-9.]+) aug ([0-9.]+)\) val_loss ([0-9.]+) val_hit@3 ([0-9.]+)", log)
    assert len(ep) == 2, log[-2000:]
    aug = [float(e[3]) for e in ep]
    assert all(np.isfinite([float(x) for e in ep for x in e[1:]])) and aug[1] < 0.3 * aug[0], ep
    assert 0.0 < float(ep[1][5]) < 1.0 and "test_metrics best epoch" in log and "test_metrics last epoch" in log
    assert (out / "checkpoint-1" / "pytorch_model.bin").exists()


def test_training_run_that_diverges_to_nan_is_loud(dev, tmp_path, monkeypatch):
    """The argparse DEFAULT ``--lambda_decay -1`` (the reference's scripts pass 0.0001) turns the time decay of ``CLtime_loss``
    (``train/train_retriever.py:48-52``) into exp(+|dt|) = inf on the UCI_13 query times: NaN loss, NaN weights after the first
    update -- in the reference too, which then validates and ranks with them in silence.  Here the epoch's loss raises a
    ``RuntimeWarning`` and the validation pass refuses the weights (range guard): no metrics, no ``*.gen`` file
    from NaN scores.  (``tools/annotation_e2e.py`` timed the evaluation of such checkpoints in round 4 without noticing.)"""
    import importlib.util
    import io
    from contextlib import redirect_stdout
    from conftest import REPO
    from rag4dyg_amd import _lib
    import main_retriever
    spec = importlib.util.spec_from_file_location("train_uci13_demo", os.path.join(REPO, "tools", "train_uci13_demo.py"))
    demo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo)
    base, ret = demo.build_workdir(str(tmp_path))
    monkeypatch.chdir(tmp_path)
    out = tmp_path / "out"
    argv = (f"--dataset UCI_13 --timestamp 12 --output_dir {out} --model_type gpt2 --model_name_or_path gpt2 "
            f"--train_data_file {base}/train.link_prediction --eval_data_file {base}/val.link_prediction "
            f"--eval_data_gt_file {ret}/val_score.retrieval --test_data_file {base}/test.link_prediction "
            f"--test_data_gt_file {ret}/test_score.retrieval --n_layer 2 --n_head 2 --n_embed 128 --block_size 512 --seed 42 --topK 5 "
            f"--train_pair_data_file {ret}/train_index.retrieval --max_steps 3 --per_gpu_train_batch_size 16 --do_train --patience 50").split()
    with pytest.warns(RuntimeWarning, match="summed training loss"), pytest.raises(_lib.R4DError, match="non-finite"):
        with redirect_stdout(io.StringIO()):
            main_retriever.main(argv)
    assert not list(tmp_path.rglob("*.gen"))


# ------------------------------------------------------------------------------------------------ dropout (training mode)
@pytest.mark.parametrize("n,p,site,base", [(4, 0.1, 0, 0), (10000, 0.1, 6, 4096), (40000, 0.5, 65535, 4 * (2 ** 32 + 5)),
                                           (1 << 20, 0.25, 9, 0)])
def test_dropout_kernel_equals_oracle_generator(dev, n, p, site, base):
    """The device's Philox-4x32-10 masks are the oracle's (bit for bit), the kept values x / (1 - p) in fp32, the residual added
    after; the keep rate is 1 - p."""
    from oracle import train_ref
    from rag4dyg_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(n)
    x, r = torch.randn(n, generator=g), torch.randn(n, generator=g)
    seed, step = 0x1234_5678_9ABC_DEF0, 2 ** 33 + 17
    keep = train_ref.philox_keep(n, p, seed, step, site, base)
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    want = np.where(keep, x.numpy() * scale, np.float32(0.0)).astype(np.float32)
    X, R = x.to(dev), r.to(dev)
    out = torch.empty_like(X)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.r4d_dropout_f32(X.data_ptr(), None, n, out.data_ptr(), p, seed, step, site, base, st), "dropout")
    assert np.array_equal(out.cpu().numpy(), want)
    _lib.check(lib.r4d_dropout_f32(X.data_ptr(), R.data_ptr(), n, R.data_ptr(), p, seed, step, site, base, st), "dropout")   # in place
    assert np.array_equal(R.cpu().numpy(), (r.numpy() + want).astype(np.float32))
    if n >= 10000:
        assert abs(keep.mean() - (1 - p)) < 4 * np.sqrt(p * (1 - p) / n)


def test_training_step_with_dropout_equals_oracle_given_the_same_masks(dev):
    """model.train(): embeddings, loss and EVERY parameter gradient of a step with dropout 0.1 / 0.15 / 0.2 (embeddings,
    attention probabilities, residual branches) against the oracle's grad-enabled forward applying the SAME masks (the
    oracle restates the generator and the element numbering) -- i.e. forward and backward use one consistent mask per site,
    scaled by 1 / (1 - p), at the places modeling_gpt2.py:153,194,212,427 put nn.Dropout; a second step draws new masks."""
    from oracle import gpt2_ref, train_ref
    from rag4dyg_amd import training
    m, args, batch, times, mask, seed = _tiny_step_inputs(dev)
    g = load_golden("g8_training_step")
    tag = "ts_tiny"
    L, H, d, V, pad, B, _ = (int(x) for x in g[tag + "_cfg"])
    eta, gamma, alpha, temp, lam = (float(x) for x in g[tag + "_hyper"])
    sd = gpt2_ref.make_state_dict(L, d, V, n_positions=1024, seed=seed, random_affine=True)
    ps = (0.1, 0.15, 0.2)
    trainer = training.EncoderTrainer(m, dropout=ps, seed=991)
    anchor, pos, neg = (b.to(dev) for b in batch[:3])
    idx = torch.cat(batch[3:6], dim=1)
    embs = []
    for step in (1, 2):
        random.seed(seed)
        aug1, aug2 = training.aug(anchor, eta, gamma, V - 1)
        emb = trainer.forward([anchor, pos, neg, aug1, aug2])
        embs.append(emb.clone())
        losses, demb = training.retriever_losses(args, emb.view(5, B, -1), times[idx[:, 0:1]], times[idx[:, 1:2]], times[idx[:, 2:3]])
        cl, au = losses[0], losses[1]
        grads = trainer.backward(demb.view(5 * B, -1))
        sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "lm_head.weight"}
        sdg["lm_head.weight"] = sdg["transformer.wte.weight"]
        drop = train_ref.PhiloxDropout(*ps, seed=991, step=step)
        r = train_ref.training_step(sdg, H, anchor.cpu(), pos.cpu(), neg.cpu(), times, idx, eta, gamma, alpha, temp, lam, V - 1, seed,
                                    with_grad=True, drop=drop)
        assert rel_err(emb.view(5, B, -1).cpu().numpy(), r["emb"].detach().numpy()) < 1e-4
        assert abs(float((cl + au).item()) / float(r["loss"].item()) - 1) < 1e-4
        r["loss"].backward()
        errs = {n: rel_err(grads[n].cpu().numpy(), sdg[n].grad.numpy()) for n in grads}
        print(f"dropout step {step}: worst element-wise (max-norm) gradient error {max(errs.values()):.2e}")
        assert max(errs.values()) < 1e-3, {n: e for n, e in errs.items() if e > 1e-3}
    assert rel_err(embs[0].cpu().numpy(), embs[1].cpu().numpy()) > 1e-2          # new masks every step
    assert rel_err(embs[0].view(5, B, -1).cpu().numpy(), g[tag + "_emb"]) > 1e-2   # and not the eval-mode forward
    m.train()                                                                    # config probabilities when the module trains
    assert training.EncoderTrainer(m)._dropout_struct().resid_p == pytest.approx(m.config.resid_pdrop)
    m.eval()
    assert training.EncoderTrainer(m)._dropout_struct() is None


# ------------------------------------------------------------------------------------------------ single backward ops
def _stream():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("rows,d,with_add", [(1, 64, False), (37, 256, True), (5003, 512, True), (3000, 768, False),
                                             (70001, 512, True), (260, 1024, True), (129, 2048, False)])
def test_layernorm_backward_equals_autograd(dev, rows, d, with_add):
    from rag4dyg_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(rows + d)
    x = torch.randn(rows, d, generator=g) * 1.7 + 0.3
    w = torch.randn(d, generator=g) * 0.5 + 1.0
    b = torch.randn(d, generator=g) * 0.1
    dy = torch.randn(rows, d, generator=g)
    add = torch.randn(rows, d, generator=g) if with_add else None
    xr = x.double().requires_grad_(True); wr = w.double().requires_grad_(True); br = b.double().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (d,), wr, br, 1e-5).backward(dy.double())
    dx_ref = xr.grad + (add.double() if with_add else 0)
    X, W, DY = x.to(dev), w.to(dev), dy.to(dev)
    dx = add.to(dev).clone() if with_add else torch.empty_like(X)             # the in-place residual form: add aliases dx
    dw, db = torch.empty(d, device=dev), torch.empty(d, device=dev)
    ws = torch.empty(lib.r4d_layernorm_bwd_workspace_bytes(rows, d), dtype=torch.uint8, device=dev)
    _lib.check(lib.r4d_layernorm_bwd_f32(X.data_ptr(), W.data_ptr(), DY.data_ptr(), dx.data_ptr() if with_add else None, rows, d, 1e-5,
                                         dx.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "ln_bwd")
    assert elementwise_err(dx.cpu().numpy(), dx_ref.numpy()) < 1
    assert rel_err(dw.cpu().numpy(), wr.grad.numpy()) < 1e-5 and rel_err(db.cpu().numpy(), br.grad.numpy()) < 1e-5


@pytest.mark.parametrize("rows,fin,fout", [(1, 4, 4), (31, 64, 192), (1000, 132, 516), (9999, 512, 512), (20011, 256, 1024),
                                           (70000, 128, 128)])
def test_weight_gradient_equals_float64(dev, rows, fin, fout):
    """dW = x^T . dy (the [K,M] x [K,N] split-K MFMA GEMM, ragged in every dimension) and db = column sums of dy."""
    from rag4dyg_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(rows)
    x, dy = torch.randn(rows, fin, generator=g), torch.randn(rows, fout, generator=g)
    X, DY = x.to(dev), dy.to(dev)
    dw, db = torch.full((fin, fout), float("nan"), device=dev), torch.full((fout,), float("nan"), device=dev)
    ws = torch.empty(lib.r4d_weight_grad_workspace_bytes(rows, fin, fout), dtype=torch.uint8, device=dev)
    for _ in range(2):                                                         # second run: same bits (fixed summation order)
        _lib.check(lib.r4d_weight_grad_f32(X.data_ptr(), DY.data_ptr(), rows, fin, fout, dw.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                           ws.numel(), _stream()), "weight_grad")
        got = (dw.cpu().clone(), db.cpu().clone())
        if _ == 0:
            first = got
    assert torch.equal(first[0], got[0]) and torch.equal(first[1], got[1])
    ref = x.double().t() @ dy.double()
    scale = (x.double().abs().t() @ dy.double().abs())                         # fp32 dot-product error scales with sum |x||dy|
    assert float(((got[0].double() - ref).abs() / scale).max()) < 2e-6
    assert float(((got[1].double() - dy.double().sum(0)).abs() / dy.double().abs().sum(0)).max()) < 2e-6


def test_gelu_and_causal_softmax_backward_equal_autograd(dev):
    from rag4dyg_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    pre, dy = torch.randn(1237, 96, generator=g) * 2.5, torch.randn(1237, 96, generator=g)
    pr = pre.double().requires_grad_(True)
    u = 0.7978845608028654 * (pr + 0.044715 * pr ** 3)
    y_ref = 0.5 * pr * (1.0 + torch.tanh(u))                                    # gelu_new, modeling_gpt2.py:25
    y_ref.backward(dy.double())
    P, DY = pre.to(dev), dy.to(dev)
    y, dx = torch.empty_like(P), torch.empty_like(P)
    _lib.check(lib.r4d_gelu_new_f32(P.data_ptr(), P.numel(), y.data_ptr(), _stream()), "gelu")
    _lib.check(lib.r4d_gelu_new_bwd_f32(P.data_ptr(), DY.data_ptr(), P.numel(), dx.data_ptr(), _stream()), "gelu_bwd")
    assert elementwise_err(y.cpu().numpy(), y_ref.detach().numpy()) < 1 and elementwise_err(dx.cpu().numpy(), pr.grad.numpy()) < 1
    # causal softmax backward on [nbh, T, ld] with the logits divided by sqrt(hd) first (modeling_gpt2.py:143-150)
    nbh, T, ld, sd = 6, 45, 128, 8.0
    logits = torch.randn(nbh, T, T, generator=g).double().requires_grad_(True)
    mask = torch.tril(torch.ones(T, T, dtype=torch.bool))
    probs = torch.softmax(torch.where(mask, logits / sd, torch.tensor(-1e4, dtype=torch.float64)), dim=-1)
    dP = torch.randn(nbh, T, T, generator=g)
    probs.backward(dP.double())
    Pd = torch.zeros(nbh, T, ld)
    Pd[:, :, :T] = torch.where(mask, probs.detach().float(), torch.zeros(()))
    dPd = torch.zeros(nbh, T, ld)
    dPd[:, :, :T] = dP
    Pd, dPd = Pd.to(dev), dPd.to(dev)
    _lib.check(lib.r4d_causal_softmax_bwd_f32(Pd.data_ptr(), dPd.data_ptr(), nbh, T, ld, sd, _stream()), "softmax_bwd")
    got = dPd.cpu()[:, :, :T]
    want = torch.where(mask, logits.grad, torch.zeros((), dtype=torch.float64))
    assert elementwise_err(got.numpy(), want.numpy()) < 1
    assert float(dPd.cpu()[:, :, T:].abs().max()) == 0.0
