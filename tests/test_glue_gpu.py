"""The head / loss glue as HIP launches (round 5: "PyTorch tensors for storage only"): the forward model's residual in the GEMM
epilogue (reference models/forward_inverse.py:27-37), th.cat((state, next_state), 1) of the inverse / reward heads (:62,78-95), the mean
of reconstructionLoss (losses/losses.py:172-181), sums of two loss scalars, and the explicit fan-out of a tensor with several
consumers — each against the torch expression it replaces, values and gradients."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_forward_model_node_matches_the_torch_expression():
    from srlz import ops
    torch.manual_seed(0)
    for B, S, A in ((7, 200, 6), (256, 200, 6), (5, 10, 4)):
        state = torch.randn(B, S, device=DEV, requires_grad=True)
        w = (0.1 * torch.randn(S, S + A, device=DEV)).requires_grad_(True)
        b = torch.randn(S, device=DEV, requires_grad=True)
        act = torch.randint(0, A, (B,), device=DEV)
        y = ops.ForwardModelFn.apply(state, act, w, b, A)
        g = torch.randn_like(y)
        y.backward(g)
        s2, w2, b2 = (t.detach().double().requires_grad_(True) for t in (state, w, b))
        onehot = torch.zeros(B, A, device=DEV, dtype=torch.float64).scatter_(1, act.view(-1, 1), 1.0)
        ref = s2 + torch.nn.functional.linear(torch.cat((s2, onehot), 1), w2, b2)
        ref.backward(g.double())
        for got, want in ((y, ref), (state.grad, s2.grad), (w.grad, w2.grad), (b.grad, b2.grad)):
            assert float((got.double() - want).abs().max()) <= 2e-5 * float(want.abs().max()), (B, S, A)


def test_cat_cols_fan_out_mean_and_scalar_sum():
    from srlz import ops
    torch.manual_seed(1)
    a = torch.randn(33, 200, device=DEV, requires_grad=True)
    b = torch.randn(33, 17, device=DEV, requires_grad=True)
    c = ops.CatColsFn.apply(a, b)
    assert torch.equal(c, torch.cat((a, b), 1))
    g = torch.randn_like(c)
    c.backward(g)
    assert torch.equal(a.grad, g[:, :200]) and torch.equal(b.grad, g[:, 200:])
    # fan-out: three consumers, one of them never differentiated through; ((g0 + g1) + g2) in take order
    t = torch.randn(64, 200, device=DEV, requires_grad=True)
    fan = ops.Fan(t, 3)
    x0, x1, x2 = fan.take(), fan.take(), fan.take()
    assert x0.data_ptr() == t.data_ptr() and fan.take() is t
    (x0 * 2.0).sum().backward(retain_graph=True)
    assert torch.equal(t.grad, torch.full_like(t, 2.0))
    t.grad = None
    w1, w2 = torch.randn_like(t), torch.randn_like(t)
    ((x0 * w1).sum() + (x2 * w2).sum()).backward()
    assert float((t.grad - (w1 + w2)).abs().max()) == 0.0 or torch.equal(t.grad, w2 + w1)
    plain = torch.randn(4, 4, device=DEV)
    assert all(p is plain for p in ops.fan_out(plain, 3))  # no gradient: no node
    # mean of squared differences: fp32(sum) / numel, and its gradient (g / numel) * 2 (a - b)
    x = torch.randn(9, 200, device=DEV, requires_grad=True)
    y = torch.randn(9, 200, device=DEV, requires_grad=True)
    m = ops.SqDiffSumFn.apply(x, y, True)
    m.backward()
    ref = ((x.detach().double() - y.detach().double()) ** 2).sum() / x.numel()
    assert abs(float(m) - float(ref)) <= 1e-6 * float(ref)
    gref = 2.0 * (x.detach().double() - y.detach().double()) / x.numel()
    assert float((x.grad.double() - gref).abs().max()) <= 1e-6 * float(gref.abs().max())
    assert torch.equal(y.grad, -x.grad)
    s = ops.SqDiffSumFn.apply(x.detach(), y.detach())
    assert float(m.detach()) == float(s / np.float32(x.numel()))  # exactly the separately rounded division
    # a + b of two loss scalars
    p, q = torch.tensor(1.25e-3, device=DEV, requires_grad=True), torch.tensor(7.5, device=DEV, requires_grad=True)
    r = ops.add_scalars(p, q)
    r.backward()
    assert float(r.detach()) == float(p.detach() + q.detach()) and float(p.grad) == 1.0 and float(q.grad) == 1.0
