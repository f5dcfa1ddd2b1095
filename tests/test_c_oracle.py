"""Pins the plain-C oracle (oracle/srlz_oracle.c via oracle/c_oracle.py): operator level against torch, step level
against the golden fixtures captured from the unmodified reference."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_util as gu
from oracle import c_oracle as O

RS = np.random.RandomState(3)


def close(a, b, rtol=2e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape
    assert np.abs(a - b).max() <= rtol * max(np.abs(b).max(), 1e-30), np.abs(a - b).max() / np.abs(b).max()


@pytest.mark.parametrize("C,K,R,st,pad,H", [(3, 8, 7, 2, 3, 20), (6, 5, 3, 1, 1, 9), (4, 4, 3, 2, 1, 11)])
def test_conv2d(C, K, R, st, pad, H):
    x, w = RS.randn(2, C, H, H).astype(np.float32), RS.randn(K, C, R, R).astype(np.float32)
    xt, wt = torch.tensor(x, dtype=torch.float64, requires_grad=True), torch.tensor(w, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(xt, wt, None, st, pad)
    dy = RS.randn(*y.shape).astype(np.float32)
    y.backward(torch.tensor(dy, dtype=torch.float64))
    close(O.conv2d_fwd(x, w, None, st, pad), y.detach().numpy())
    dx, dw, _ = O.conv2d_bwd(x, w, dy, st, pad)
    close(dx, xt.grad.numpy())
    close(dw, wt.grad.numpy())


@pytest.mark.parametrize("C,K,R,H", [(5, 4, 3, 6), (6, 3, 4, 7)])
def test_conv_transpose2d(C, K, R, H):
    x, w, b = RS.randn(2, C, H, H).astype(np.float32), RS.randn(C, K, R, R).astype(np.float32), RS.randn(K).astype(np.float32)
    xt, wt, bt = (torch.tensor(t, dtype=torch.float64, requires_grad=True) for t in (x, w, b))
    y = F.conv_transpose2d(xt, wt, bt, 2)
    dy = RS.randn(*y.shape).astype(np.float32)
    y.backward(torch.tensor(dy, dtype=torch.float64))
    close(O.convT2d_fwd(x, w, b, 2), y.detach().numpy())
    dx, dw, db = O.convT2d_bwd(x, w, dy, 2)
    close(dx, xt.grad.numpy())
    close(dw, wt.grad.numpy())
    close(db, bt.grad.numpy())


def test_batchnorm_relu_pool_linear_losses_adam():
    x = (RS.randn(3, 4, 9, 9) * 1.5 + 0.3).astype(np.float32)
    g, b = (RS.rand(4) + 0.5).astype(np.float32), RS.randn(4).astype(np.float32)
    rm, rv = np.zeros(4, np.float32), np.ones(4, np.float32)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    gt, bt = torch.tensor(g, dtype=torch.float64, requires_grad=True), torch.tensor(b, dtype=torch.float64, requires_grad=True)
    rmt, rvt = torch.zeros(4, dtype=torch.float64), torch.ones(4, dtype=torch.float64)
    z = F.batch_norm(xt, rmt, rvt, gt, bt, True, 0.1, 1e-5)
    p, idx = F.max_pool2d(F.relu(z), 3, 2, 1, return_indices=True)
    dp = RS.randn(*p.shape).astype(np.float32)
    p.backward(torch.tensor(dp, dtype=torch.float64))
    zc, mean, invstd = O.bn_train_fwd(x, g, b, rm, rv)
    close(zc, z.detach().numpy())
    close(rm, rmt.numpy())
    close(rv, rvt.numpy())
    rc = O.relu_fwd(zc)
    pc, ic = O.maxpool_fwd(rc, 1)
    close(pc, p.detach().numpy())
    assert np.array_equal(ic, idx.numpy())
    dz = O.relu_bwd(zc, O.maxpool_bwd(dp, ic, rc.shape))
    dx, dg, db = O.bn_train_bwd(x, dz, g, mean, invstd)
    close(dx, xt.grad.numpy(), 5e-5)
    close(dg, gt.grad.numpy(), 5e-5)
    close(db, bt.grad.numpy(), 5e-5)
    close(O.bn_eval_fwd(x, g, b, rm, rv), F.batch_norm(xt.detach(), rmt, rvt, gt.detach(), bt.detach(), False, 0.1, 1e-5).numpy())
    # linear
    a, w, bb = RS.randn(5, 7).astype(np.float32), RS.randn(3, 7).astype(np.float32), RS.randn(3).astype(np.float32)
    at, wt, bbt = (torch.tensor(t, dtype=torch.float64, requires_grad=True) for t in (a, w, bb))
    yl = F.linear(at, wt, bbt)
    dyl = RS.randn(5, 3).astype(np.float32)
    yl.backward(torch.tensor(dyl, dtype=torch.float64))
    close(O.linear_fwd(a, w, bb), yl.detach().numpy())
    dxa, dwl, dbl = O.linear_bwd(a, w, dyl)
    close(dxa, at.grad.numpy()); close(dwl, wt.grad.numpy()); close(dbl, bbt.grad.numpy())
    # losses (KATs of the reference's free functions, tests/golden/loss_kats.npz)
    k = gu.load("loss_kats")
    n = k["in/a"].size
    assert abs(O.sqdiff_sum(k["in/a"], k["in/b"]) / n - float(k["reconstruction"])) < 1e-6
    assert abs(2.0 * (O.kl_sum(k["in/mu"], k["in/lv"]) + O.kl_sum(k["in/nmu"], k["in/nlv"])) - float(k["kl_beta2"])) < 1e-4
    ce, _ = O.cross_entropy(k["in/logits"], k["in/act"].reshape(-1))
    assert abs(2.0 * ce - float(k["inverse_w2"])) < 1e-6
    # Adam vs torch.optim.Adam
    p0 = RS.randn(50).astype(np.float32)
    pt = torch.tensor(p0.copy(), requires_grad=True)
    opt = torch.optim.Adam([pt], lr=5e-3)
    pc_, m, v = p0.copy(), np.zeros(50, np.float32), np.zeros(50, np.float32)
    for step in (1, 2, 3):
        gr = RS.randn(50).astype(np.float32)
        pt.grad = torch.tensor(gr)
        opt.step()
        O.adam_step(pc_, gr, m, v, 5e-3, step)
    close(pc_, pt.detach().numpy(), 1e-6)


def test_c_autoencoder_step_matches_reference_golden():
    """The composed C oracle reproduces the reference's B=2 auto-encoder step (loss, states, reconstruction, grads)."""
    from test_oracle_golden import build
    g = gu.load("step_ae_b2")
    model = build(["autoencoder"])
    sd = {k: v.detach().numpy() for k, v in model.state_dict().items()}
    obs, next_obs, _ = gu.golden_inputs(2, 3, 6, seed=1234)
    out = O.ae_train_step(sd, obs, next_obs)
    assert abs(out["total"] - float(g["loss/total"])) <= 1e-5 * float(g["loss/total"])
    gu.check_digest(out["states"], g, "states", rtol=5e-5)
    gu.check_digest(out["decoded"], g, "decoded", rtol=5e-5)
    gu.check_digest(out["next_decoded"], g, "next_decoded", rtol=5e-5)
    # gradients of the reference carry fp32 tie-break noise (tests/test_step_gpu.py docstring): norm-level agreement
    for k, grad in out["grads"].items():
        if k.endswith(("decoder_conv.0.bias", "decoder_conv.3.bias", "decoder_conv.6.bias", "decoder_conv.9.bias")):
            continue
        l2 = float(g["grad/" + k + "/l2"])
        assert abs(np.sqrt((grad.astype(np.float64) ** 2).sum()) - l2) <= 2e-2 * l2, k
    for k in [f for f in g.files if f.startswith("bn/") and "running" in f]:
        np.testing.assert_allclose(out["sd"][k[3:]], g[k], rtol=2e-5, atol=1e-7)
