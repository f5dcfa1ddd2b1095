"""conv3x3(64, 64) stride 1 pad 1 as Winograd F(2x2, 3x3) (csrc/wino.hip) — conv2 of the encoder, /root/reference/models/models.py:54 —
through the C ABI against fp64 torch on the host (the oracle's operator) and against the direct fp32 implicit GEMM (conv64_fwd_kernel).

Tolerances: the transforms add a couple of fp32 roundings per operand on top of the direct chain's; against fp64 F.conv2d every output
element is held to 2e-5 of the layer's output scale at the toy sizes (the bar the direct kernel is held to) and the measured error is
printed; the BatchNorm partial records are held to the sums of the kernel's own output."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def rel_err(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)


@pytest.fixture(scope="module")
def C():
    from srlz import _cabi
    assert torch.cuda.is_available()
    return _cabi


# (images, size, groups, bias): one patch row per image .. ragged last tiles .. two groups whose tile ranges differ .. the step's size
WINO = [(1, 4, 1, False), (3, 56, 1, False), (2, 8, 2, True), (5, 10, 1, True), (6, 56, 2, False), (7, 14, 1, False), (4, 6, 2, False),
        (512, 56, 2, False)]


@pytest.mark.parametrize("n,h,groups,with_bias", WINO)
def test_conv64_wino_forward(C, n, h, groups, with_bias):
    g = torch.Generator().manual_seed(h * 7 + n)
    x = torch.randn(n, 64, h, h, generator=g)
    x[:, :, 0, :] += 1.5  # (border rows / columns carry a signal of their own: a wrong padding mask shows)
    x[:, :, :, -1] -= 1.5
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    b = torch.randn(64, generator=g) if with_bias else None
    ref = F.conv2d(x.double(), w.double(), b.double() if with_bias else None, stride=1, padding=1)
    d = C.Conv64Desc(n, h, h, h, h, 3, 1, 1, 0, groups)
    assert C.conv64_wino_supported(d) == 1
    st = C.stream()
    xd, wd = nhwc(x).to(DEV), w.to(DEV)
    bd = b.to(DEV) if with_bias else None
    up = torch.full((2, C.conv64_wino_packed_floats()), float("nan"), device=DEV)
    C.conv64_wino_pack_weights(C.ptr(wd), C.ptr(up[0]), C.ptr(up[1]), st)
    rows = C.conv64_wino_tiles(d)
    y = torch.full((n, h, h, 64), float("nan"), device=DEV)
    stats = torch.full((rows, 128), float("nan"), device=DEV)
    C.conv64_wino_fwd(C.ptr(xd), C.ptr(up[0]), C.ptr(bd), C.ptr(y), C.ptr(stats), None, d, st)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all() and torch.isfinite(stats).all()
    err = rel_err(nchw(y), ref)
    # the direct fp32 chain on the same inputs
    packs = torch.empty(2, C.conv64_packed_floats(), device=DEV)
    C.conv64_pack_weights(C.ptr(wd), C.ptr(packs[0]), C.ptr(packs[1]), d, st)
    y2 = torch.empty_like(y)
    C.conv64_fwd(C.ptr(xd), C.ptr(packs[0]), C.ptr(bd), C.ptr(y2), None, None, d, st)
    torch.cuda.synchronize()
    err_direct = rel_err(nchw(y2), ref)
    print("wino n=%d h=%d: max error / output scale %.2e (direct fp32 chain %.2e)" % (n, h, err, err_direct))
    assert err < 2e-5, err
    # BatchNorm partial records: per group, sum and sum of squares of the kernel's own output
    per = n // groups
    st64 = stats.double().view(groups, rows // groups, 128).sum(1).cpu()
    for gi in range(groups):
        yg = y[gi * per:(gi + 1) * per].double().reshape(-1, 64).cpu()
        assert (st64[gi, :64] - yg.sum(0)).abs().max().item() <= 1e-5 * yg.abs().sum(0).max().item()
        assert (st64[gi, 64:] - (yg ** 2).sum(0)).abs().max().item() <= 1e-5 * (yg ** 2).sum(0).max().item()
    # without the statistics output; and a second launch, bit for bit
    y3 = torch.empty_like(y)
    C.conv64_wino_fwd(C.ptr(xd), C.ptr(up[0]), C.ptr(bd), C.ptr(y3), None, None, d, st)
    torch.cuda.synchronize()
    assert torch.equal(y3, y)


def test_conv64_wino_does_not_depend_on_batching(C):
    """The arithmetic of a patch does not depend on the tile or launch it falls into: images alone, batched, or batched in two BatchNorm
    groups give the same bits (what lets the two frames of a step, models/learner.py:392-393, run as one launch)."""
    g = torch.Generator().manual_seed(5)
    x = nhwc(torch.randn(6, 64, 12, 12, generator=g)).to(DEV)
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(DEV)
    st = C.stream()
    up = torch.empty(2, C.conv64_wino_packed_floats(), device=DEV)
    C.conv64_wino_pack_weights(C.ptr(w), C.ptr(up[0]), None, st)

    def run(xs, groups):
        d = C.Conv64Desc(xs.shape[0], 12, 12, 12, 12, 3, 1, 1, 0, groups)
        y = torch.empty_like(xs)
        stats = torch.empty(C.conv64_wino_tiles(d), 128, device=DEV)
        C.conv64_wino_fwd(C.ptr(xs), C.ptr(up[0]), None, C.ptr(y), C.ptr(stats), None, d, st)
        torch.cuda.synchronize()
        return y, stats

    y_all, _ = run(x, 1)
    y_two, s_two = run(x, 2)
    assert torch.equal(y_all, y_two)
    for i in range(6):
        y_i, _ = run(x[i:i + 1].contiguous(), 1)
        assert torch.equal(y_i[0], y_all[i])
    # a group's records are those of the group run alone
    y_a, s_a = run(x[:3].contiguous(), 1)
    y_b, s_b = run(x[3:].contiguous(), 1)
    assert torch.equal(s_two, torch.cat([s_a, s_b]))


def test_conv64_wino_rejects_what_it_cannot_run(C):
    for d in (C.Conv64Desc(2, 27, 27, 14, 14, 3, 2, 1, 0, 1), C.Conv64Desc(2, 13, 13, 27, 27, 3, 2, 0, 1, 1),
              C.Conv64Desc(2, 9, 9, 9, 9, 3, 1, 1, 0, 1), C.Conv64Desc(3, 8, 8, 8, 8, 3, 1, 1, 0, 2)):
        assert C.conv64_wino_supported(d) == 0


@pytest.mark.parametrize("n,h,groups", [(2, 8, 1), (3, 56, 1), (6, 20, 2), (512, 56, 2)])
def test_conv64_wino_data_gradient(C, n, h, groups):
    """srlz_conv64_wino_bwd_data: d(loss)/dx of conv3x3 s1 p1 — the same kernel on G g' G^T of the flipped, transposed weights — against
    fp64 autograd (every element) and the direct data-gradient kernel."""
    g = torch.Generator().manual_seed(h + 3 * n)
    x = torch.randn(n, 64, h, h, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    dy = torch.randn(n, 64, h, h, generator=g)
    xr = x.double().requires_grad_(True)
    F.conv2d(xr, w.double(), None, stride=1, padding=1).backward(dy.double())
    d = C.Conv64Desc(n, h, h, h, h, 3, 1, 1, 0, groups)
    st = C.stream()
    wd, dyd = w.to(DEV), nhwc(dy).to(DEV)
    up = torch.empty(2, C.conv64_wino_packed_floats(), device=DEV)
    C.conv64_wino_pack_weights(C.ptr(wd), None, C.ptr(up[1]), st)
    dx = torch.full((n, h, h, 64), float("nan"), device=DEV)
    C.conv64_wino_bwd_data(C.ptr(dyd), C.ptr(up[1]), C.ptr(dx), d, st)
    torch.cuda.synchronize()
    err = rel_err(nchw(dx), xr.grad)
    packs = torch.empty(2, C.conv64_packed_floats(), device=DEV)
    C.conv64_pack_weights(C.ptr(wd), C.ptr(packs[0]), C.ptr(packs[1]), d, st)
    dx2 = torch.empty_like(dx)
    C.conv64_bwd_data(C.ptr(dyd), C.ptr(packs[1]), C.ptr(dx2), None, d, st)
    torch.cuda.synchronize()
    print("wino dgrad n=%d h=%d: max error / scale %.2e (direct fp32 chain %.2e)" % (n, h, err, rel_err(nchw(dx2), xr.grad)))
    assert err < 2e-5, err


@pytest.mark.parametrize("n,h,groups,zero", [(4, 24, 2, False), (3, 112, 1, True), (512, 112, 2, False)])
def test_conv64_wino_data_gradient_with_the_pooled_blocks_sums(C, n, h, groups, zero):
    """srlz_conv64_wino_bwd_data_pool_sums against srlz_conv64_bwd_data_pool_sums (the direct kernel's pooled-block epilogue, itself held
    to fp64 autograd by test_pool_block_bn_backward_sums_from_the_next_convs_data_gradient): same d(pooled) up to the Winograd roundings,
    the two BatchNorm-backward sums / dgamma / dbeta after srlz_bn_bwd_finalize_partials at 2e-5 — with channels whose BatchNorm scale is
    exactly 0 and tiny (the companion launch)."""
    g = torch.Generator().manual_seed(h + n)
    hp = (h + 2 - 3) // 2 + 1
    y1 = torch.randn(n, h, h, 64, generator=g) * 1.3 + 0.2
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3
    if zero:
        gamma[5], beta[5] = 0.0, 0.25
        gamma[40], beta[40] = 0.0, -0.1
        gamma[7], beta[7] = 1e-6, 0.3
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(DEV)
    dy = torch.randn(n, hp, hp, 64, generator=g).to(DEV)
    st = C.stream()
    yd = y1.to(DEV)
    per = n // groups
    # BatchNorm records of the pooled block (training statistics per group) and its forward: pooled map + argmax
    bnp = torch.empty(groups * 256, device=DEV)
    for gi in range(groups):
        ys = yd[gi * per:(gi + 1) * per].reshape(-1, 64).double()
        mean, var = ys.mean(0), ys.var(0, unbiased=False)
        inv = 1.0 / torch.sqrt(var + 1e-5)
        sc = gamma.to(DEV).double() * inv
        bnp[gi * 256:(gi + 1) * 256] = torch.cat([mean, inv, sc, beta.to(DEV).double() - mean * sc]).float()
    pd = C.PoolDesc(n, h, h, hp, hp, 1, 0, groups)
    pooled = torch.empty(n, hp, hp, 64, device=DEV)
    argmax = torch.empty(n, hp, hp, 64, dtype=torch.uint8, device=DEV)
    C.bn_relu_pool_fwd(C.ptr(yd), C.ptr(bnp), C.ptr(pooled), C.ptr(argmax), pd, st)
    d = C.Conv64Desc(n, hp, hp, hp, hp, 3, 1, 1, 0, groups)
    up = torch.empty(2, C.conv64_wino_packed_floats(), device=DEV)
    C.conv64_wino_pack_weights(C.ptr(w), None, C.ptr(up[1]), st)
    packs = torch.empty(2, C.conv64_packed_floats(), device=DEV)
    C.conv64_pack_weights(C.ptr(w), C.ptr(packs[0]), C.ptr(packs[1]), d, st)
    rows_w, rows_d = C.conv64_wino_bwd_data_rows(d), C.conv64_bwd_data_tiles(d)
    dx_w, dx_d = torch.full((n, hp, hp, 64), float("nan"), device=DEV), torch.empty(n, hp, hp, 64, device=DEV)
    part_w, part_d = torch.full((rows_w, 128), float("nan"), device=DEV), torch.empty(rows_d, 128, device=DEV)
    C.conv64_wino_bwd_data_pool_sums(C.ptr(dy), C.ptr(up[1]), C.ptr(dx_w), C.ptr(pooled), C.ptr(bnp), C.ptr(yd), C.ptr(argmax), pd,
                                     C.ptr(part_w), d, st)
    C.conv64_bwd_data_pool_sums(C.ptr(dy), C.ptr(packs[1]), C.ptr(dx_d), C.ptr(pooled), C.ptr(bnp), C.ptr(yd), C.ptr(argmax), pd,
                                C.ptr(part_d), d, st)
    torch.cuda.synchronize()
    assert torch.isfinite(dx_w).all() and torch.isfinite(part_w).all()
    assert rel_err(dx_w, dx_d) < 5e-6
    dx_plain = torch.empty_like(dx_w)
    C.conv64_wino_bwd_data(C.ptr(dy), C.ptr(up[1]), C.ptr(dx_plain), d, st)
    torch.cuda.synchronize()
    assert torch.equal(dx_plain, dx_w)  # the epilogue changes nothing about the gradient itself
    nb = C.bn_bwd_workspace(0)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    outs = []
    for part, rows in ((part_w, rows_w), (part_d, rows_d)):
        sums, dg, db = torch.empty(128 * groups, device=DEV), torch.empty(64, device=DEV), torch.empty(64, device=DEV)
        C.bn_bwd_finalize_partials(C.ptr(part), rows, groups, C.ptr(sums), C.ptr(dg), C.ptr(db), C.ptr(ws), nb, st)
        outs.append((sums, dg, db))
    torch.cuda.synchronize()
    for got, want in zip(outs[0], outs[1]):
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 2e-5 * scale, float((got - want).abs().max()) / scale


@pytest.mark.parametrize("n,h", [(1, 4), (2, 8), (3, 56), (5, 10), (7, 14), (64, 56), (512, 56)])
def test_conv64_wino_weight_gradient(C, n, h):
    """srlz_conv64_wino_bwd_weight: dW = G^T [sum over patches (A dY A^T) .* (B^T x B)] G against fp64 autograd (every element; the
    sums run over n * h * h positions — up to 1.6 M at the step's size, held to north_star's 1e-4 there) and the direct weight gradient;
    deterministic (two launches, bit for bit)."""
    g = torch.Generator().manual_seed(11 * h + n)
    x = torch.randn(n, 64, h, h, generator=g)
    x[:, :, 0, :] += 1.0
    x[:, :, :, -1] -= 1.0
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    dy = torch.randn(n, 64, h, h, generator=g)
    wr = w.double().requires_grad_(True)
    F.conv2d(x.double(), wr, None, stride=1, padding=1).backward(dy.double())
    d = C.Conv64Desc(n, h, h, h, h, 3, 1, 1, 0, 2 if n % 2 == 0 else 1)
    st = C.stream()
    xd, dyd = nhwc(x).to(DEV), nhwc(dy).to(DEV)
    nb = C.conv64_wino_bwd_weight_workspace(d)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    dw = torch.full((64, 64, 3, 3), float("nan"), device=DEV)
    C.conv64_wino_bwd_weight(C.ptr(xd), C.ptr(dyd), C.ptr(dw), C.ptr(ws), nb, d, st)
    dw2 = torch.empty_like(dw)
    ws.fill_(255)
    C.conv64_wino_bwd_weight(C.ptr(xd), C.ptr(dyd), C.ptr(dw2), C.ptr(ws), nb, d, st)
    torch.cuda.synchronize()
    assert torch.isfinite(dw).all() and torch.equal(dw, dw2)
    nb2 = C.conv64_bwd_weight_workspace(d)
    ws2 = torch.empty(nb2, dtype=torch.uint8, device=DEV)
    dwd = torch.empty_like(dw)
    C.conv64_bwd_weight(C.ptr(xd), C.ptr(dyd), C.ptr(dwd), None, None, None, C.ptr(ws2), nb2, d, st)
    torch.cuda.synchronize()
    err, err_direct = rel_err(dw, wr.grad), rel_err(dwd, wr.grad)
    print("wino wgrad n=%d h=%d: max error / scale %.2e (direct weight gradient %.2e)" % (n, h, err, err_direct))
    assert err < (2e-5 if n < 512 else 1e-4), err


@pytest.mark.parametrize("n,h,groups", [(2, 8, 1), (3, 10, 1), (6, 56, 2), (12, 14, 6)])
def test_conv64_wino_forward_with_the_fused_bn_relu_operand(C, n, h, groups):
    """srlz_conv64_wino_fwd with x_bnp: the tensor is the RAW output of the previous convolution, the layer's input relu(batchnorm(raw))
    per BatchNorm group — applied in the landing, the zero padding re-imposed behind it (the second convolution of a ResNet block,
    models/triplet.py:16) — against fp64 torch and the direct kernel's fused operand; BatchNorm partial records included."""
    g = torch.Generator().manual_seed(17 * h + n)
    x = torch.randn(n, 64, h, h, generator=g) * 1.5
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    per = n // groups
    scale, shift = torch.randn(groups, 64, generator=g), torch.randn(groups, 64, generator=g) * 0.5   # (negative scales included)
    act = torch.cat([F.relu(x[gi * per:(gi + 1) * per].double() * scale[gi].double().view(1, 64, 1, 1) + shift[gi].double().view(1, 64, 1, 1))
                     for gi in range(groups)])
    ref = F.conv2d(act, w.double(), None, stride=1, padding=1)
    bnp = torch.zeros(groups, 256)
    bnp[:, 128:192], bnp[:, 192:256] = scale, shift
    d = C.Conv64Desc(n, h, h, h, h, 3, 1, 1, 0, groups)
    st = C.stream()
    xd, wd, bd = nhwc(x).to(DEV), w.to(DEV), bnp.to(DEV)
    up = torch.empty(2, C.conv64_wino_packed_floats(), device=DEV)
    C.conv64_wino_pack_weights(C.ptr(wd), C.ptr(up[0]), None, st)
    rows = C.conv64_wino_tiles(d)
    y = torch.full((n, h, h, 64), float("nan"), device=DEV)
    stats = torch.full((rows, 128), float("nan"), device=DEV)
    C.conv64_wino_fwd(C.ptr(xd), C.ptr(up[0]), None, C.ptr(y), C.ptr(stats), C.ptr(bd), d, st)
    packs = torch.empty(2, C.conv64_packed_floats(), device=DEV)
    C.conv64_pack_weights(C.ptr(wd), C.ptr(packs[0]), C.ptr(packs[1]), d, st)
    y2 = torch.empty_like(y)
    C.conv64_fwd(C.ptr(xd), C.ptr(packs[0]), None, C.ptr(y2), None, C.ptr(bd), d, st)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all() and torch.isfinite(stats).all()
    err = rel_err(nchw(y), ref)
    print("wino fused n=%d h=%d: max error / output scale %.2e (direct fused kernel %.2e)" % (n, h, err, rel_err(nchw(y2), ref)))
    assert err < 2e-5, err
    st64 = stats.double().view(groups, rows // groups, 128).sum(1).cpu()
    for gi in range(groups):
        yg = y[gi * per:(gi + 1) * per].double().reshape(-1, 64).cpu()
        assert (st64[gi, :64] - yg.sum(0)).abs().max().item() <= 1e-5 * yg.abs().sum(0).max().item()
        assert (st64[gi, 64:] - (yg ** 2).sum(0)).abs().max().item() <= 1e-5 * (yg ** 2).sum(0).max().item()
