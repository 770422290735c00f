"""GPU test of the retriever training step's FORWARD half (staged) against the reference's own step
(tests/golden/g8_training_step.npz: five forwards, CLtime_loss + alpha * info_nce, computed by the reference on CPU)."""
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
    r = training.training_step_forward(args, m, batch, T(g[tag + "_times"]), training.mask_correlated_samples(B))
    assert np.array_equal(r["aug1"].cpu().numpy(), g[tag + "_aug1"]) and np.array_equal(r["aug2"].cpu().numpy(), g[tag + "_aug2"])
    emb = r["embeddings"].cpu().numpy()
    assert rel_err(emb, g[tag + "_emb"]) < 1e-4 and elementwise_err(emb, g[tag + "_emb"]) < 1
    got = np.array([r["cl_loss"].item(), r["aug_loss"].item(), r["loss"].item()])
    print(f"{tag}: losses {got} vs reference {g[tag + '_losses']}, embeddings max-norm err {rel_err(emb, g[tag + '_emb']):.2e}")
    assert np.abs(got / g[tag + "_losses"] - 1).max() < 1e-4, (got, g[tag + "_losses"])


@pytest.mark.parametrize("tag", ["ts_tiny", "ts_cfg2"])
def test_training_step_gradients_equal_reference_autograd(dev, tag):
    """Backward pass on the HIP kernels against the REFERENCE's autograd (G8: one train_epoch iteration, dropout 0): the
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
    leaf = emb.view(5, B, -1).detach().requires_grad_(True)
    t = T(g[tag + "_times"])
    with torch.enable_grad():
        cl = training.CLtime_loss(args, leaf[0], leaf[1], leaf[2], t[idx[:, 0:1]], t[idx[:, 1:2]], t[idx[:, 2:3]])
        au = alpha * training.info_nce(args, leaf[3], leaf[4], temp, B, training.mask_correlated_samples(B))
        (cl + au).backward()
    grads = trainer.backward(leaf.grad.view(5 * B, -1))
    names = [str(n) for n in g[tag + "_grad_names"]]
    got = np.array([grads["transformer.wte.weight" if n == "lm_head.weight" else n].double().norm().item() for n in names])
    worst = np.abs(got / g[tag + "_grad_norms"] - 1).max()
    print(f"{tag}: worst relative error of a parameter's gradient norm {worst:.2e}")
    assert worst < 1e-3, (worst, [n for n, a, b in zip(names, got, g[tag + "_grad_norms"]) if abs(a / b - 1) > 1e-3])
    assert rel_err(grads["transformer.ln_f.weight"].cpu().numpy(), g[tag + "_grad_lnf_w"]) < 1e-3
    assert rel_err(grads["transformer.h.0.attn.c_attn.bias"].cpu().numpy(), g[tag + "_grad_cattn_b0"]) < 1e-3
    assert rel_err(grads["transformer.wte.weight"][:8].cpu().numpy(), g[tag + "_grad_wte_rows"]) < 1e-3
    # every tensor element-wise against the oracle's autograd
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != "lm_head.weight"}
    sdg["lm_head.weight"] = sdg["transformer.wte.weight"]
    r = train_ref.training_step(sdg, H, anchor.cpu(), pos.cpu(), neg.cpu(), g[tag + "_times"], idx, eta, gamma, alpha, temp, lam,
                                V - 1, seed, with_grad=True)
    r["loss"].backward()
    errs = {n: rel_err(grads[n].cpu().numpy(), sdg[n].grad.numpy()) for n in grads}
    bad = {n: e for n, e in errs.items() if e > 1e-3}
    print(f"{tag}: worst element-wise (max-norm) gradient error {max(errs.values()):.2e}")
    assert not bad, bad
