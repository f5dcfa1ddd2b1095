"""Kernel-level parity (GPU): every C-ABI entry point against the CPU oracle on the same seeded inputs.

Oracle = torch CPU fp32/fp64 functional ops (the third-party library the reference's arithmetic lives in) — see
oracle/torch_twin.py header.  Tolerance: fp32, 1e-4 relative (BASELINE.json north_star); most ops hold 1e-5.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def rel_err(got, ref):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    scale = ref.abs().max().item()
    return (got - ref).abs().max().item() / max(scale, 1e-30)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


@pytest.fixture(scope="module")
def C():
    from srlz import _cabi
    assert torch.cuda.is_available()
    return _cabi


def out_size(hi, s, p, t):
    return (hi - 1) * s - 2 * p + 3 if t else (hi + 2 * p - 3) // s + 1


CONV64 = [(3, 56, 1, 1, 0), (3, 27, 2, 1, 0), (5, 6, 2, 0, 1), (3, 13, 2, 0, 1), (2, 27, 2, 0, 1), (2, 55, 2, 0, 1),
          (1, 9, 1, 1, 0),
          (1, 80, 1, 1, 0)]  # 128 + 2 * 82 + 2 = 294 rows per tile: the row table's second batch of passes (rowtab_passes = 32)


# ... and the four big layers at the headline step's size (round 6): N = 512 images in two BatchNorm groups — thousands of tiles on the
# XCD walk, the weight-gradient rings (stride 1 and stride 2) and the gather weight gradient with hundreds of workgroups of split-K
# partials + conv64_wgrad_reduce, conv64_gather_pipe_kernel (conv3: 450 tiles per group) — every output element against fp64
# F.conv2d / F.conv_transpose2d autograd on the host (reference models/models.py:54,59,66-78 under one loss.backward(), learner.py:489)
CONV64_FULL = [(512, 56, 1, 1, 0, 2), (512, 27, 2, 1, 0, 2), (512, 27, 2, 0, 1, 2), (512, 55, 2, 0, 1, 2)]


@pytest.mark.parametrize("n,hi,s,p,t,groups", [c + (1,) for c in CONV64] + CONV64_FULL)
def test_conv64_forward_backward(C, n, hi, s, p, t, groups):
    g = torch.Generator().manual_seed(hi * 13 + s)
    # fp32 chains over 128 .. 12 000 positions against fp64: 2e-5 at the toy sizes; the full-size rows sum 1.6 .. 6.3 M positions per
    # weight-gradient element and are held to north_star's 1e-4 (measured: printed below)
    tol = 2e-5 if n < 512 else 1e-4
    ho = out_size(hi, s, p, t)
    x = torch.randn(n, 64, hi, hi, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    b = torch.randn(64, generator=g) if t else None
    dy = torch.randn(n, 64, ho, ho, generator=g)
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if t else None
    if t:
        yr = F.conv_transpose2d(xr, wr, br, stride=s, padding=p)
    else:
        yr = F.conv2d(xr, wr, None, stride=s, padding=p)
    yr.backward(dy.double())

    d = C.Conv64Desc(n, hi, hi, ho, ho, 3, s, p, t, groups)
    st = C.stream()
    xd, wd, dyd = nhwc(x).to(DEV), w.to(DEV), nhwc(dy).to(DEV)
    bd = b.to(DEV) if t else None
    packs = torch.empty(2, C.conv64_packed_floats(), device=DEV)
    C.conv64_pack_weights(C.ptr(wd), C.ptr(packs[0]), C.ptr(packs[1]), d, st)
    y = torch.full((n, ho, ho, 64), float("nan"), device=DEV)
    ntiles = C.conv64_fwd_tiles(d)
    if n >= 512 and not t and s == 2:  # conv3 at the step's size runs the pipelined persistent kernel
        assert C.conv64_gather_pipe_supported(d, 0) == 1
    stats = torch.empty(ntiles, 128, device=DEV)
    C.conv64_fwd(C.ptr(xd), C.ptr(packs[0]), C.ptr(bd), C.ptr(y), C.ptr(stats), None, d, st)
    torch.cuda.synchronize()
    assert rel_err(nchw(y), yr) < tol
    # BatchNorm partial sums
    s_tot = stats.double().sum(0).cpu()
    yr_flat = yr.detach().permute(1, 0, 2, 3).reshape(64, -1)
    assert rel_err(s_tot[:64], yr_flat.sum(1)) < 1e-4 or (s_tot[:64] - yr_flat.sum(1)).abs().max() < 1e-2
    assert rel_err(s_tot[64:], (yr_flat ** 2).sum(1)) < tol

    dx = torch.full((n, hi, hi, 64), float("nan"), device=DEV)
    C.conv64_bwd_data(C.ptr(dyd), C.ptr(packs[1]), C.ptr(dx), None, d, st)
    torch.cuda.synchronize()
    e_dx = rel_err(nchw(dx), xr.grad)
    assert e_dx < tol

    nbytes = C.conv64_bwd_weight_workspace(d)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    dw = torch.full((64, 64, 3, 3), float("nan"), device=DEV)
    db = torch.full((64,), float("nan"), device=DEV)
    C.conv64_bwd_weight(C.ptr(xd), C.ptr(dyd), C.ptr(dw), C.ptr(db), None, None, C.ptr(ws), nbytes, d, st)
    torch.cuda.synchronize()
    e_dw = rel_err(dw, wr.grad)
    assert e_dw < tol, e_dw
    ref_db = dy.double().sum((0, 2, 3))
    assert rel_err(db, ref_db) < tol
    if n >= 512:
        print("conv64 n=%d hi=%d s=%d t=%d: dx %.2e dw %.2e" % (n, hi, s, t, e_dx, e_dw))


def test_conv64_at_the_32_bit_offset_limit(C):
    """The tile tables keep 28-bit pixel indices and 32-bit float offsets per BatchNorm group (build_program / launch_fwd): a group of the
    widest layer (111 x 111 x 64) may hold floor((2^32 - 1) / (111 * 111 * 64)) = 5446 images.  Until round 6 that bound was guarded by a
    check and by nothing else.  Here the last ConvTranspose block's data gradient (dy [n,111,111,64] -> dx [n,55,55,64]: the staged rows'
    offsets reach 2^32 - 2.6 MB) and forward (x -> y: the destination side) run at n = 5440 images in ONE group, and images from the
    start, the middle and the very end are compared with fp64 F.conv_transpose2d autograd; one image beyond the bound is refused with
    SRLZ_ERR_BAD_DESC instead of wrapping around.  Reference: models/models.py:78 (ConvTranspose2d(64, 64, 3, stride 2))."""
    hi, ho = 55, 111
    limit = (2 ** 32 - 1) // (ho * ho * 64)
    n = 5440
    assert n <= limit < n + 16
    g = torch.Generator(device=DEV).manual_seed(11)
    w = torch.randn(64, 64, 3, 3, generator=g, device=DEV) * 0.05
    b = torch.randn(64, generator=g, device=DEV)
    st = C.stream()
    d = C.Conv64Desc(n, hi, hi, ho, ho, 3, 2, 0, 1, 1)
    packs = torch.empty(2, C.conv64_packed_floats(), device=DEV)
    C.conv64_pack_weights(C.ptr(w), C.ptr(packs[0]), C.ptr(packs[1]), d, st)
    x = torch.randn(n, hi, hi, 64, generator=g, device=DEV)                    # 4.2 GB
    y = torch.empty(n, ho, ho, 64, device=DEV)                                 # 17.2 GB
    stats = torch.empty(C.conv64_fwd_tiles(d), 128, device=DEV)
    C.conv64_fwd(C.ptr(x), C.ptr(packs[0]), C.ptr(b), C.ptr(y), C.ptr(stats), None, d, st)
    torch.cuda.synchronize()
    pick = [0, 1, n // 2, n - 2, n - 1]
    wr, br = w.double().cpu(), b.double().cpu()
    for i in pick:
        ref = F.conv_transpose2d(nchw(x[i:i + 1]).double().cpu(), wr, br, stride=2)
        assert rel_err(nchw(y[i:i + 1]), ref) < 2e-5, i
    # the data gradient reads the 17 GB tensor through the row table's offsets; dy = y (any values will do)
    dx = torch.full((n, hi, hi, 64), float("nan"), device=DEV)
    C.conv64_bwd_data(C.ptr(y), C.ptr(packs[1]), C.ptr(dx), None, d, st)
    torch.cuda.synchronize()
    for i in pick:
        a = nchw(x[i:i + 1]).double().cpu().requires_grad_(True)
        F.conv_transpose2d(a, wr, None, stride=2).backward(nchw(y[i:i + 1]).double().cpu())
        assert rel_err(nchw(dx[i:i + 1]), a.grad) < 2e-5, i
    assert torch.isfinite(dx).all()
    # one image too many: refused, nothing launched (the buffers would be too small for it)
    from srlz._cabi import SrlzError
    too_many = C.Conv64Desc(limit + 1, hi, hi, ho, ho, 3, 2, 0, 1, 1)
    with pytest.raises(SrlzError):
        C.conv64_bwd_data(C.ptr(y), C.ptr(packs[1]), C.ptr(dx), None, too_many, st)
    del x, y, dx
    torch.cuda.empty_cache()


def test_conv64_deterministic(C):
    """Two runs of the weight gradient are bit-identical (fixed-order split-K reduction; learner.py:62)."""
    n, hi = 4, 27
    g = torch.Generator().manual_seed(5)
    d = C.Conv64Desc(n, hi, hi, 55, 55, 3, 2, 0, 1)
    x = torch.randn(n, hi, hi, 64, generator=g).to(DEV)
    dy = torch.randn(n, 55, 55, 64, generator=g).to(DEV)
    nbytes = C.conv64_bwd_weight_workspace(d)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    outs = []
    for _ in range(2):
        dw = torch.empty(64, 64, 3, 3, device=DEV)
        C.conv64_bwd_weight(C.ptr(x), C.ptr(dy), C.ptr(dw), None, None, None, C.ptr(ws), nbytes, d, C.stream())
        outs.append(dw.cpu())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("n,hi,s,p,t,groups", [(300, 27, 2, 1, 0, 2), (146, 27, 2, 1, 0, 1), (512, 27, 2, 1, 0, 2), (120, 31, 2, 1, 0, 1),
                                                (170, 13, 2, 0, 1, 1), (360, 13, 2, 0, 1, 2)])
def test_gather_pipe_kernel_is_the_synchronous_kernel(C, n, hi, s, p, t, groups):
    """conv64_gather_pipe_kernel (round 5: the plain stride-2 gather programs — conv3's forward, a ConvTranspose's data gradient —
    persistent and software-pipelined) against conv64_fwd_kernel<4, false>, which still takes the same program when a bias is
    given: same tiles, same accumulation order -> the outputs are bit-identical (a zero bias adds +0); the per-tile BatchNorm
    partials have the same tile geometry and agree to summation order.  One and two BatchNorm groups, more tiles than workgroups
    (528 / 900 tiles on 512 persistent workgroups), a ragged grid (31 -> 16) and a ConvTranspose's data gradient included.  The
    kernel takes a program from 256 tiles PER GROUP on — a per-group criterion, so that a group alone and the batched pair of a
    step run the same kernel; below that the synchronous kernel is faster (bs = 32: 33 us against 51).
    Reference: models/models.py:59 (conv3x3 stride 2) / :66-78 (ConvTranspose2d(64, 64, 3, stride 2))."""
    g = torch.Generator().manual_seed(n * 7 + hi)
    ho = out_size(hi, s, p, t)
    d = C.Conv64Desc(n, hi, hi, ho, ho, 3, s, p, t, groups)
    small = C.Conv64Desc(8 * groups, hi, hi, ho, ho, 3, s, p, t, groups)
    assert C.conv64_gather_pipe_supported(small, 1 if t else 0) == 0  # few tiles per group: the synchronous kernel
    st = C.stream()
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(DEV)
    packs = torch.empty(2, C.conv64_packed_floats(), device=DEV)
    C.conv64_pack_weights(C.ptr(w), C.ptr(packs[0]), C.ptr(packs[1]), d, st)
    zero = torch.zeros(64, device=DEV)
    if not t:  # forward of a stride-2 convolution: plain operand, statistics
        assert C.conv64_gather_pipe_supported(d, 0) == 1
        x = torch.randn(n, hi, hi, 64, generator=g).to(DEV)
        tiles = C.conv64_fwd_tiles(d)
        outs = []
        for bias in (None, zero):
            y = torch.full((n, ho, ho, 64), float("nan"), device=DEV)
            stats = torch.full((tiles, 128), float("nan"), device=DEV)
            C.conv64_fwd(C.ptr(x), C.ptr(packs[0]), C.ptr(bias), C.ptr(y), C.ptr(stats), None, d, st)
            torch.cuda.synchronize()
            outs.append((y, stats))
        assert torch.equal(outs[0][0], outs[1][0])
        a, b = outs[0][1].double(), outs[1][1].double()
        assert torch.isfinite(a).all() and float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
        y2 = torch.full((n, ho, ho, 64), float("nan"), device=DEV)  # eval mode: no statistics
        C.conv64_fwd(C.ptr(x), C.ptr(packs[0]), None, C.ptr(y2), None, None, d, st)
        torch.cuda.synchronize()
        assert torch.equal(y2, outs[0][0])
        # deterministic
        y3, st3 = torch.empty_like(y2), torch.empty_like(outs[0][1])
        C.conv64_fwd(C.ptr(x), C.ptr(packs[0]), None, C.ptr(y3), C.ptr(st3), None, d, st)
        torch.cuda.synchronize()
        assert torch.equal(y3, y2) and torch.equal(st3, outs[0][1])
    else:  # data gradient of a ConvTranspose: against fp64 torch (no second route exists for it)
        assert C.conv64_gather_pipe_supported(d, 1) == 1
        dy = torch.randn(n, ho, ho, 64, generator=g).to(DEV)
        dx = torch.full((n, hi, hi, 64), float("nan"), device=DEV)
        C.conv64_bwd_data(C.ptr(dy), C.ptr(packs[1]), C.ptr(dx), None, d, st)
        torch.cuda.synchronize()
        xr = torch.zeros(n, 64, hi, hi, dtype=torch.float64, requires_grad=True)
        F.conv_transpose2d(xr, w.double().cpu(), None, stride=s, padding=p).backward(nchw(dy).double().cpu())
        assert rel_err(nchw(dx), xr.grad) < 2e-5


@pytest.mark.parametrize("n,c,h", [(2, 3, 224), (1, 6, 224), (2, 3, 64), (1, 9, 96)])
def test_conv1(C, n, c, h):
    g = torch.Generator().manual_seed(c * 100 + h)
    x = torch.randn(n, c, h, h, generator=g)
    w = torch.randn(64, c, 7, 7, generator=g) * 0.1
    hf = (h + 6 - 7) // 2 + 1
    dy = torch.randn(n, 64, hf, hf, generator=g)
    wr = w.double().requires_grad_(True)
    yr = F.conv2d(x.double(), wr, None, stride=2, padding=3)
    yr.backward(dy.double())
    d = C.SkinnyDesc(n, c, h, h, hf, hf, 0)
    st = C.stream()
    xd, wd, dyd = x.to(DEV), w.to(DEV), nhwc(dy).to(DEV)
    y = torch.full((n, hf, hf, 64), float("nan"), device=DEV)
    stats = torch.empty(C.skinny_tiles(d), 128, device=DEV)
    C.conv1_fwd(C.ptr(xd), C.ptr(wd), C.ptr(y), C.ptr(stats), d, st)
    torch.cuda.synchronize()
    assert rel_err(nchw(y), yr) < 2e-5
    s_tot = stats.double().sum(0).cpu()
    yr_flat = yr.detach().permute(1, 0, 2, 3).reshape(64, -1)
    assert rel_err(s_tot[64:], (yr_flat ** 2).sum(1)) < 2e-5
    assert (s_tot[:64] - yr_flat.sum(1)).abs().max() < 1e-3 * yr_flat.abs().sum(1).max()
    nbytes = C.skinny_bwd_weight_workspace(d)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    dw = torch.full((64, c, 7, 7), float("nan"), device=DEV)
    C.conv1_bwd_weight(C.ptr(xd), C.ptr(dyd), C.ptr(dw), C.ptr(ws), nbytes, d, st)
    torch.cuda.synchronize()
    assert rel_err(dw, wr.grad) < 2e-5


@pytest.mark.parametrize("n,c,hf", [(2, 3, 111), (1, 6, 111), (3, 3, 20), (1, 9, 30), (2, 3, 1), (1, 6, 16), (5, 3, 33)])
def test_convT_out(C, n, c, hf):
    g = torch.Generator().manual_seed(c * 10 + hf)
    himg = (hf - 1) * 2 + 4
    x = torch.randn(n, 64, hf, hf, generator=g)
    w = torch.randn(64, c, 4, 4, generator=g) * 0.1
    b = torch.randn(c, generator=g)
    dy = torch.randn(n, c, himg, himg, generator=g)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, br, stride=2)
    yr.backward(dy.double())
    d = C.SkinnyDesc(n, c, himg, himg, hf, hf, 1)
    st = C.stream()
    xd, wd, bd, dyd = nhwc(x).to(DEV), w.to(DEV), b.to(DEV), dy.to(DEV)
    y = torch.full((n, c, himg, himg), float("nan"), device=DEV)
    C.convT_out_fwd(C.ptr(xd), C.ptr(wd), C.ptr(bd), C.ptr(y), None, d, st)
    torch.cuda.synchronize()
    assert rel_err(y, yr) < 2e-5
    dx = torch.full((n, hf, hf, 64), float("nan"), device=DEV)
    C.convT_out_bwd_data(C.ptr(dyd), C.ptr(wd), C.ptr(dx), None, None, None, d, st)
    torch.cuda.synchronize()
    assert rel_err(nchw(dx), xr.grad) < 2e-5
    nbytes = C.skinny_bwd_weight_workspace(d)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    dw = torch.full((64, c, 4, 4), float("nan"), device=DEV)
    db = torch.full((c,), float("nan"), device=DEV)
    C.convT_out_bwd_weight(C.ptr(xd), C.ptr(dyd), C.ptr(dw), C.ptr(db), None, C.ptr(ws), nbytes, d, st)
    torch.cuda.synchronize()
    assert rel_err(dw, wr.grad) < 2e-5
    assert rel_err(db, br.grad) < 2e-5


def _bn_setup(g, shape):
    y = torch.randn(*shape, generator=g) * 1.7 + 0.3
    gamma = torch.rand(64, generator=g) + 0.5
    beta = torch.randn(64, generator=g) * 0.2
    rm = torch.randn(64, generator=g) * 0.1
    rv = torch.rand(64, generator=g) + 0.5
    return y, gamma, beta, rm, rv


def _partials(y_nhwc, rows_per=100):
    """Per-tile partial sums as a conv epilogue would produce them (any tiling sums to the same statistics)."""
    flat = y_nhwc.reshape(-1, 64).double()
    chunks = torch.split(flat, rows_per)
    return torch.stack([torch.cat((c.sum(0), (c * c).sum(0))) for c in chunks]).float()


@pytest.mark.parametrize("n,h,pad,out_nchw,training", [(2, 112, 1, 0, 1), (3, 56, 0, 0, 1), (4, 14, 0, 1, 1),
                                                       (2, 56, 0, 0, 0), (2, 14, 0, 1, 0)])
def test_bn_relu_pool(C, n, h, pad, out_nchw, training):
    g = torch.Generator().manual_seed(h + pad)
    y, gamma, beta, rm, rv = _bn_setup(g, (n, 64, h, h))
    hp = (h + 2 * pad - 3) // 2 + 1
    dp = torch.randn(n, 64, hp, hp, generator=g)
    yr = y.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm_r, rv_r = rm.double().clone(), rv.double().clone()
    z = F.batch_norm(yr, rm_r, rv_r, gr, br, bool(training), 0.1, 1e-5)
    pr = F.max_pool2d(F.relu(z), 3, 2, pad)
    pr.backward(dp.double())

    st = C.stream()
    yd = nhwc(y).to(DEV)
    gd, bd, rmd, rvd = gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV)
    bnp = torch.empty(256, device=DEV)
    if training:
        parts = _partials(nhwc(y)).to(DEV)
        bstat = torch.empty(128, device=DEV)
        fws = torch.empty(C.bn_bwd_workspace(0), dtype=torch.uint8, device=DEV)
        C.bn_finalize(C.ptr(parts), parts.shape[0], 1, n * h * h, C.ptr(gd), C.ptr(bd), 1e-5, 0.1, 1, C.ptr(rmd),
                      C.ptr(rvd), None, C.ptr(bnp), C.ptr(bstat), C.ptr(fws), fws.numel(), st)
        torch.cuda.synchronize()
        assert rel_err(rmd, rm_r) < 1e-5 and rel_err(rvd, rv_r) < 1e-5
    else:
        C.bn_eval_params(C.ptr(gd), C.ptr(bd), C.ptr(rmd), C.ptr(rvd), 1e-5, C.ptr(bnp), st)
    d = C.PoolDesc(n, h, h, hp, hp, pad, out_nchw)
    pooled = torch.full((n, 64, hp, hp) if out_nchw else (n, hp, hp, 64), float("nan"), device=DEV)
    arg = torch.empty(n, hp, hp, 64, dtype=torch.uint8, device=DEV)
    C.bn_relu_pool_fwd(C.ptr(yd), C.ptr(bnp), C.ptr(pooled), C.ptr(arg), d, st)
    torch.cuda.synchronize()
    got = pooled if out_nchw else nchw(pooled)
    assert rel_err(got, pr) < 1e-5
    dpd = (dp if out_nchw else nhwc(dp)).to(DEV)
    dy = torch.full((n, h, h, 64), float("nan"), device=DEV)
    dgm, dbt = torch.empty(64, device=DEV), torch.empty(64, device=DEV)
    nbytes = C.bn_bwd_workspace(0)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    C.bn_relu_pool_bwd(C.ptr(yd), C.ptr(bnp), C.ptr(arg), C.ptr(dpd), C.ptr(pooled), C.ptr(dy), C.ptr(dgm), C.ptr(dbt), training,
                       C.ptr(ws), nbytes, d, st)
    torch.cuda.synchronize()
    assert rel_err(nchw(dy), yr.grad) < 5e-5
    assert rel_err(dgm, gr.grad) < 5e-5 and rel_err(dbt, br.grad) < 5e-5


@pytest.mark.parametrize("n,h,training", [(2, 111, 1), (3, 13, 1), (2, 27, 0)])
def test_bn_relu(C, n, h, training):
    g = torch.Generator().manual_seed(h)
    y, gamma, beta, rm, rv = _bn_setup(g, (n, 64, h, h))
    da = torch.randn(n, 64, h, h, generator=g)
    yr = y.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ar = F.relu(F.batch_norm(yr, rm.double().clone(), rv.double().clone(), gr, br, bool(training), 0.1, 1e-5))
    ar.backward(da.double())
    st = C.stream()
    yd, dad = nhwc(y).to(DEV), nhwc(da).to(DEV)
    gd, bd, rmd, rvd = gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV)
    bnp = torch.empty(256, device=DEV)
    if training:
        parts = _partials(nhwc(y)).to(DEV)
        fws = torch.empty(C.bn_bwd_workspace(0), dtype=torch.uint8, device=DEV)
        C.bn_finalize(C.ptr(parts), parts.shape[0], 1, n * h * h, C.ptr(gd), C.ptr(bd), 1e-5, 0.1, 1, C.ptr(rmd),
                      C.ptr(rvd), None, C.ptr(bnp), None, C.ptr(fws), fws.numel(), st)
    else:
        C.bn_eval_params(C.ptr(gd), C.ptr(bd), C.ptr(rmd), C.ptr(rvd), 1e-5, C.ptr(bnp), st)
    a = torch.full((n, h, h, 64), float("nan"), device=DEV)
    C.bn_relu_fwd(C.ptr(yd), C.ptr(bnp), C.ptr(a), n * h * h, st)
    torch.cuda.synchronize()
    assert rel_err(nchw(a), ar) < 1e-5
    dy = torch.full((n, h, h, 64), float("nan"), device=DEV)
    dgm, dbt = torch.empty(64, device=DEV), torch.empty(64, device=DEV)
    nbytes = C.bn_bwd_workspace(0)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    C.bn_relu_bwd(C.ptr(yd), C.ptr(bnp), C.ptr(dad), C.ptr(dy), C.ptr(dgm), C.ptr(dbt), training, C.ptr(ws), nbytes,
                  n * h * h, 1, st)
    torch.cuda.synchronize()
    assert rel_err(nchw(dy), yr.grad) < 5e-5
    assert rel_err(dgm, gr.grad) < 5e-5 and rel_err(dbt, br.grad) < 5e-5


def test_bn_replay_and_repeat(C):
    g = torch.Generator().manual_seed(3)
    stat = torch.randn(128, generator=g).abs()
    rm, rv = torch.randn(64, generator=g), torch.rand(64, generator=g)
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    stat_d = stat.to(DEV)
    C.bn_replay(C.ptr(stat_d), 0.1, C.ptr(rmd), C.ptr(rvd), C.stream())
    torch.cuda.synchronize()
    assert rel_err(rmd, 0.9 * rm + 0.1 * stat[:64]) < 1e-6
    assert rel_err(rvd, 0.9 * rv + 0.1 * stat[64:]) < 1e-6


@pytest.mark.parametrize("M,N,K,relu", [(32, 200, 2304, 0), (7, 2304, 200, 0), (5, 6, 400, 0), (64, 128, 400, 1),
                                        (3, 200, 206, 0), (130, 70, 33, 1), (256, 200, 2304, 0), (256, 2304, 200, 0),
                                        (256, 128, 1000, 1)])
def test_linear(C, M, N, K, relu):
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.1
    b = torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.linear(xr, wr, br)
    if relu:
        yr = F.relu(yr)
    yr.backward(dy.double())
    st = C.stream()
    xd, wd, bd, dyd = x.to(DEV), w.to(DEV), b.to(DEV), dy.to(DEV)
    y = torch.full((M, N), float("nan"), device=DEV)
    nb = C.linear_workspace(M, N, K)
    lws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    C.linear_fwd(C.ptr(xd), C.ptr(wd), C.ptr(bd), C.ptr(y), M, N, K, relu, C.ptr(lws), nb, st)
    torch.cuda.synchronize()
    assert rel_err(y, yr) < 2e-5
    if relu:
        C.relu_bwd_inplace(C.ptr(y), C.ptr(dyd), M * N, st)
    dx = torch.full((M, K), float("nan"), device=DEV)
    C.linear_bwd_data(C.ptr(dyd), C.ptr(wd), C.ptr(dx), M, N, K, C.ptr(lws), nb, st)
    dw = torch.full((N, K), float("nan"), device=DEV)
    db = torch.full((N,), float("nan"), device=DEV)
    C.linear_bwd_weight(C.ptr(dyd), C.ptr(xd), C.ptr(dw), C.ptr(db), M, N, K, None, 0, st)
    torch.cuda.synchronize()
    assert rel_err(dx, xr.grad) < 2e-5 and rel_err(dw, wr.grad) < 2e-5 and rel_err(db, br.grad) < 2e-5


def test_layout_seams(C):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(5, 64, 6, 6, generator=g)
    xd = x.to(DEV)
    y = torch.empty(5, 6, 6, 64, device=DEV)
    C.nchw_to_nhwc(C.ptr(xd), C.ptr(y), 5, 64, 6, 6, C.stream())
    back = torch.empty(5, 64, 6, 6, device=DEV)
    C.nhwc_to_nchw(C.ptr(y), C.ptr(back), 5, 64, 6, 6, C.stream())
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), nhwc(x)) and torch.equal(back.cpu(), x)


@pytest.mark.parametrize("n", [1, 7, 4096, 2 * 3 * 224 * 224 + 3])
def test_loss_reductions(C, n):
    g = torch.Generator().manual_seed(n)
    a, b = torch.randn(n + 4, generator=g)[:n].contiguous(), torch.randn(n, generator=g)
    st = C.stream()
    ad, bd = a.to(DEV), b.to(DEV)
    out = torch.empty((), device=DEV)
    nbytes = C.reduce_workspace(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    C.sqdiff_sum(C.ptr(ad), C.ptr(bd), n, C.ptr(out), C.ptr(ws), nbytes, st)
    ref = ((a.double() - b.double()) ** 2).sum()
    assert abs(out.item() - ref.item()) <= 2e-6 * abs(ref.item())
    coef = torch.tensor(0.37, device=DEV)
    da = torch.empty(n, device=DEV)
    C.sqdiff_grad(C.ptr(ad), C.ptr(bd), C.ptr(coef), 2.0, C.ptr(da), n, st)
    assert rel_err(da, 0.37 * 2.0 * (a.double() - b.double())) < 1e-6
    lv = (b * 0.3).contiguous()
    lvd = lv.to(DEV)
    C.kl_sum(C.ptr(ad), C.ptr(lvd), n, C.ptr(out), C.ptr(ws), nbytes, st)
    ref = -0.5 * (1 + lv.double() - a.double() ** 2 - lv.double().exp()).sum()
    assert abs(out.item() - ref.item()) <= 2e-6 * max(abs(ref.item()), 1.0)
    dmu, dlv = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    C.kl_grad(C.ptr(ad), C.ptr(lvd), C.ptr(coef), 1.0, C.ptr(dmu), C.ptr(dlv), n, st)
    assert rel_err(dmu, 0.37 * a.double()) < 1e-6
    assert rel_err(dlv, 0.37 * 0.5 * (lv.double().exp() - 1)) < 1e-5


def test_reparam_ce_onehot(C):
    g = torch.Generator().manual_seed(21)
    B, S, A = 9, 200, 6
    mu, lv, eps = torch.randn(B, S, generator=g), torch.randn(B, S, generator=g) * 0.4, torch.randn(B, S, generator=g)
    dz = torch.randn(B, S, generator=g)
    st = C.stream()
    mud, lvd, epsd, dzd = mu.to(DEV), lv.to(DEV), eps.to(DEV), dz.to(DEV)
    z = torch.empty(B, S, device=DEV)
    C.reparam_fwd(C.ptr(mud), C.ptr(lvd), C.ptr(epsd), C.ptr(z), B * S, st)
    assert rel_err(z, eps.double() * (0.5 * lv.double()).exp() + mu.double()) < 1e-6
    dmu, dlv = torch.empty(B, S, device=DEV), torch.empty(B, S, device=DEV)
    C.reparam_bwd(C.ptr(dzd), C.ptr(lvd), C.ptr(epsd), C.ptr(dmu), C.ptr(dlv), B * S, st)
    assert rel_err(dmu, dz) < 1e-7
    assert rel_err(dlv, dz.double() * eps.double() * 0.5 * (0.5 * lv.double()).exp()) < 1e-6
    logits = torch.randn(B, A, generator=g) * 2
    tgt = torch.randint(0, A, (B,), generator=g)
    lr = logits.double().requires_grad_(True)
    ref = F.cross_entropy(lr, tgt)
    ref.backward()
    out = torch.empty((), device=DEV)
    dl = torch.empty(B, A, device=DEV)
    logits_d, tgt_d = logits.to(DEV), tgt.to(DEV)  # keep the device copies alive while the kernel runs
    C.cross_entropy(C.ptr(logits_d), C.ptr(tgt_d), B, A, C.ptr(out), C.ptr(dl), st)
    assert abs(out.item() - ref.item()) < 1e-6 * max(1.0, abs(ref.item()))
    assert rel_err(dl, lr.grad) < 1e-5
    cat = torch.empty(B, S + A, device=DEV)
    C.concat_onehot(C.ptr(mud), C.ptr(tgt_d), C.ptr(cat), B, S, A, st)
    ref_cat = torch.cat((mu, F.one_hot(tgt, A).float()), 1)
    assert torch.equal(cat.cpu(), ref_cat)


def test_adam(C):
    g = torch.Generator().manual_seed(33)
    n = 10007
    p0 = torch.randn(n, generator=g)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=5e-3)
    pd = p0.to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 4):
        gr = torch.randn(n, generator=g) * (10.0 ** -step)
        pr.grad = gr.clone()
        opt.step()
        g_d = (gr * 4).to(DEV)
        C.adam_step(C.ptr(pd), C.ptr(g_d), C.ptr(m), C.ptr(v), n, 5e-3, 0.9, 0.999, 1e-8, step, 0.25, C.stream())
    assert rel_err(pd, pr) < 1e-6
    # the moments too: (1 - beta) must be torch's double-evaluated scalar (1.f - 0.999f is 1.3e-5 off)
    st = opt.state[pr]
    assert rel_err(m, st["exp_avg"]) < 5e-7 and rel_err(v, st["exp_avg_sq"]) < 5e-7


def test_fused_bn_relu_operand(C):
    """x_bnp: the layer input is relu(x*scale+shift), applied inside the operand load (never materialised)."""
    g = torch.Generator().manual_seed(77)
    n, hi = 2, 13
    ho = (hi - 1) * 2 + 3
    x = torch.randn(n, 64, hi, hi, generator=g)
    scale, shift = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3
    bnp = torch.cat((torch.zeros(64), torch.ones(64), scale, shift))
    w, b = torch.randn(64, 64, 3, 3, generator=g) * 0.05, torch.randn(64, generator=g)
    dy = torch.randn(n, 64, ho, ho, generator=g)
    a = F.relu(x.double() * scale.double().view(1, 64, 1, 1) + shift.double().view(1, 64, 1, 1))
    wr = w.double().requires_grad_(True)
    yr = F.conv_transpose2d(a, wr, b.double(), stride=2)
    yr.backward(dy.double())
    d = C.Conv64Desc(n, hi, hi, ho, ho, 3, 2, 0, 1)
    st = C.stream()
    xd, wd, bd, dyd, bnpd = nhwc(x).to(DEV), w.to(DEV), b.to(DEV), nhwc(dy).to(DEV), bnp.to(DEV)
    packs = torch.empty(2, C.conv64_packed_floats(), device=DEV)
    C.conv64_pack_weights(C.ptr(wd), C.ptr(packs[0]), C.ptr(packs[1]), d, st)
    y = torch.full((n, ho, ho, 64), float("nan"), device=DEV)
    C.conv64_fwd(C.ptr(xd), C.ptr(packs[0]), C.ptr(bd), C.ptr(y), None, C.ptr(bnpd), d, st)
    torch.cuda.synchronize()
    assert rel_err(nchw(y), yr) < 2e-5
    nbytes = C.conv64_bwd_weight_workspace(d)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    dw = torch.full((64, 64, 3, 3), float("nan"), device=DEV)
    C.conv64_bwd_weight(C.ptr(xd), C.ptr(dyd), C.ptr(dw), None, C.ptr(bnpd), None, C.ptr(ws), nbytes, d, st)
    torch.cuda.synchronize()
    assert rel_err(dw, wr.grad) < 2e-5
    # last layer (64 -> 3, 4x4 s2)
    hf = 20
    himg = (hf - 1) * 2 + 4
    xf = torch.randn(n, 64, hf, hf, generator=g)
    wt, bt = torch.randn(64, 3, 4, 4, generator=g) * 0.1, torch.randn(3, generator=g)
    dimg = torch.randn(n, 3, himg, himg, generator=g)
    af = F.relu(xf.double() * scale.double().view(1, 64, 1, 1) + shift.double().view(1, 64, 1, 1))
    wtr = wt.double().requires_grad_(True)
    out = F.conv_transpose2d(af, wtr, bt.double(), stride=2)
    out.backward(dimg.double())
    d1 = C.SkinnyDesc(n, 3, himg, himg, hf, hf, 1)
    xfd, wtd, btd, dimgd = nhwc(xf).to(DEV), wt.to(DEV), bt.to(DEV), dimg.to(DEV)
    img = torch.full((n, 3, himg, himg), float("nan"), device=DEV)
    C.convT_out_fwd(C.ptr(xfd), C.ptr(wtd), C.ptr(btd), C.ptr(img), C.ptr(bnpd), d1, st)
    torch.cuda.synchronize()
    assert rel_err(img, out) < 2e-5
    nb1 = C.skinny_bwd_weight_workspace(d1)
    ws1 = torch.empty(nb1, dtype=torch.uint8, device=DEV)
    dwt, dbt = torch.full((64, 3, 4, 4), float("nan"), device=DEV), torch.empty(3, device=DEV)
    C.convT_out_bwd_weight(C.ptr(xfd), C.ptr(dimgd), C.ptr(dwt), C.ptr(dbt), C.ptr(bnpd), C.ptr(ws1), nb1, d1, st)
    torch.cuda.synchronize()
    assert rel_err(dwt, wtr.grad) < 2e-5


@pytest.mark.parametrize("n,c", [(3, 3), (2, 6), (2, 9)])
def test_normalize_u8_bit_exact(C, n, c):
    """uint8 frames -> normalised fp32 [N,C,W,H]: bit-identical to the host arithmetic of the reference's loader."""
    from preprocessing.utils import preprocessInput
    from srlz import ops
    rs = np.random.RandomState(n + c)
    frames = rs.randint(0, 256, (n, 224, 224, c)).astype(np.uint8)
    ref = np.stack([np.dstack([preprocessInput(f[..., 3 * v:3 * v + 3].astype(np.float32)) for v in range(c // 3)])
                    .transpose(2, 1, 0) for f in frames])
    got = ops.normalize_u8(torch.from_numpy(frames).to(DEV)).cpu().numpy()
    assert got.shape == ref.shape and np.array_equal(got, ref)


@pytest.mark.parametrize("n,c,h,training", [(2, 3, 224, True), (1, 6, 224, True), (3, 3, 50, True), (2, 3, 64, False)])
def test_encoder_input_block_fused_backward(C, n, c, h, training):
    """srlz_conv1_bwd_weight_fused (dy rebuilt in the operand load) == conv1 -> BN -> ReLU -> MaxPool(3,2,1) in fp64."""
    from srlz import ops
    g = torch.Generator().manual_seed(7 * h + c)
    x = torch.randn(n, c, h, h, generator=g)
    w = torch.randn(64, c, 7, 7, generator=g) * 0.1
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    rm, rv = torch.randn(64, generator=g) * 0.1, torch.rand(64, generator=g) + 0.5
    hf = (h + 6 - 7) // 2 + 1
    hp = (hf + 2 - 3) // 2 + 1
    dp = torch.randn(n, 64, hp, hp, generator=g)

    wr, gr, br = (t.double().requires_grad_(True) for t in (w, gamma, beta))
    yr = F.conv2d(x.double(), wr, None, stride=2, padding=3)
    zr = F.batch_norm(yr, rm.double().clone(), rv.double().clone(), gr, br, training, 0.1, 1e-5)
    pr = F.max_pool2d(F.relu(zr), 3, 2, 1)
    pr.backward(dp.double())

    xd = x.to(DEV)
    wd, gd, bd = (t.to(DEV).requires_grad_(True) for t in (w, gamma, beta))
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    pooled, y = ops.EncInFn.apply(xd, wd, gd, bd, rmd, rvd, training, 1, None)
    assert rel_err(nchw(y), yr) < 2e-5 and rel_err(nchw(pooled), pr) < 2e-5
    pooled.backward(nhwc(dp).to(DEV))
    torch.cuda.synchronize()
    assert rel_err(wd.grad, wr.grad) < 5e-5
    assert rel_err(gd.grad, gr.grad) < 5e-5 and rel_err(bd.grad, br.grad) < 5e-5


@pytest.mark.parametrize("n,hi,s,p,t,training", [(2, 13, 2, 0, 1, 1), (2, 27, 2, 0, 1, 1), (3, 20, 1, 1, 0, 1),
                                                 (2, 13, 2, 0, 1, 0)])
def test_fused_bn_backward_operand(C, n, hi, s, p, t, training):
    """srlz_bn_bwd_operand: conv -> BatchNorm -> ReLU backward with d(loss)/dy rebuilt inside the data-gradient and
    weight-gradient kernels' operand load, against fp64 autograd through conv + batch_norm + relu."""
    g = torch.Generator().manual_seed(1000 + hi + s)
    ho = out_size(hi, s, p, t)
    x = torch.randn(n, 64, hi, hi, generator=g)
    w, b = torch.randn(64, 64, 3, 3, generator=g) * 0.05, torch.randn(64, generator=g) * 0.1
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    rm, rv = torch.randn(64, generator=g) * 0.1, torch.rand(64, generator=g) + 0.5
    da = torch.randn(n, 64, ho, ho, generator=g)
    xr, wr, br = (v.double().requires_grad_(True) for v in (x, w, b))
    gr, ber = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    conv = F.conv_transpose2d if t else F.conv2d
    yr = conv(xr, wr, br, stride=s, padding=p)
    yr.retain_grad()
    ar = F.relu(F.batch_norm(yr, rm.double().clone(), rv.double().clone(), gr, ber, bool(training), 0.1, 1e-5))
    ar.backward(da.double())

    st = C.stream()
    d = C.Conv64Desc(n, hi, hi, ho, ho, 3, s, p, t)
    xd, wd, bd, dad = nhwc(x).to(DEV), w.to(DEV), b.to(DEV), nhwc(da).to(DEV)
    packs = torch.empty(2, C.conv64_packed_floats(), device=DEV)
    C.conv64_pack_weights(C.ptr(wd), C.ptr(packs[0]), C.ptr(packs[1]), d, st)
    y = torch.empty(n, ho, ho, 64, device=DEV)
    stats = torch.empty(C.conv64_fwd_tiles(d), 128, device=DEV)
    C.conv64_fwd(C.ptr(xd), C.ptr(packs[0]), C.ptr(bd), C.ptr(y), C.ptr(stats), None, d, st)
    gd, bed, rmd, rvd = gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV)
    bnp = torch.empty(256, device=DEV)
    nbn = C.bn_bwd_workspace(0)
    bws = torch.empty(nbn, dtype=torch.uint8, device=DEV)
    if training:
        bstat = torch.empty(128, device=DEV)
        C.bn_finalize(C.ptr(stats), stats.shape[0], 1, n * ho * ho, C.ptr(gd), C.ptr(bed), 1e-5, 0.1, 1, C.ptr(rmd), C.ptr(rvd),
                      None, C.ptr(bnp), C.ptr(bstat), C.ptr(bws), nbn, st)
    else:
        C.bn_eval_params(C.ptr(gd), C.ptr(bed), C.ptr(rmd), C.ptr(rvd), 1e-5, C.ptr(bnp), st)
    sums, dgm, dbt = torch.empty(128, device=DEV), torch.empty(64, device=DEV), torch.empty(64, device=DEV)
    C.bn_relu_bwd_sums(C.ptr(y), C.ptr(bnp), C.ptr(dad), C.ptr(sums), C.ptr(dgm), C.ptr(dbt), C.ptr(bws), nbn, n * ho * ho, 1, st)
    dy_out = torch.full((n, ho, ho, 64), float("nan"), device=DEV)
    op = C.BnBwdOperand(y.data_ptr(), bnp.data_ptr(), sums.data_ptr(), n * ho * ho, training, dy_out.data_ptr())
    dx = torch.full((n, hi, hi, 64), float("nan"), device=DEV)
    C.conv64_bwd_data(C.ptr(dad), C.ptr(packs[1]), C.ptr(dx), op, d, st)
    nbytes = C.conv64_bwd_weight_workspace(d)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    dw, db = torch.full((64, 64, 3, 3), float("nan"), device=DEV), torch.full((64,), float("nan"), device=DEV)
    op2 = C.BnBwdOperand(y.data_ptr(), bnp.data_ptr(), sums.data_ptr(), n * ho * ho, training, None)
    C.conv64_bwd_weight(C.ptr(xd), C.ptr(dad), C.ptr(dw), C.ptr(db), None, op2, C.ptr(ws), nbytes, d, st)
    dw2 = torch.full((64, 64, 3, 3), float("nan"), device=DEV)
    C.conv64_bwd_weight(C.ptr(xd), C.ptr(dy_out), C.ptr(dw2), None, None, None, C.ptr(ws), nbytes, d, st)
    torch.cuda.synchronize()
    assert rel_err(dgm, gr.grad) < 5e-5 and rel_err(dbt, ber.grad) < 5e-5
    assert rel_err(nchw(dx), xr.grad) < 5e-5
    assert rel_err(nchw(dy_out), yr.grad) < 5e-5  # by-product: d(loss)/dy, every element written exactly once
    assert rel_err(dw, wr.grad) < 5e-5 and rel_err(dw2, wr.grad) < 5e-5
    if training:  # the bias gradient of a convolution followed by train-mode BatchNorm is identically zero
        assert db.abs().max().item() < 1e-3 * da.abs().sum().item() / 64
    else:
        assert rel_err(db, br.grad) < 5e-5


def test_mask_columns_and_param_norms(C):
    """srlz_mask_columns (detachSplit) and srlz_param_norms / _grad (l1Loss, l2Loss) against torch, incl. edge cases:
    empty kept range, full range, odd tensor lengths, an all-zero tensor under the 2-norm."""
    from srlz import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(7, 100, generator=g)
    for lo, hi in ((0, 0), (0, 100), (20, 80), (99, 100)):
        xd = x.to(DEV).requires_grad_(True)
        y = ops.MaskColumnsFn.apply(xd, lo, hi)
        ref = torch.zeros_like(x)
        ref[:, lo:hi] = x[:, lo:hi]
        assert torch.equal(y.cpu(), ref)
        y.backward(torch.ones_like(y))
        gref = torch.zeros_like(x)
        gref[:, lo:hi] = 1.0
        assert torch.equal(xd.grad.cpu(), gref)
    shapes = [(64, 3, 7, 7), (64,), (200, 2304), (1,), (13, 5), (33,)]
    params = [torch.randn(*s, generator=g) for s in shapes]
    params[3].zero_()
    for mode in (0, 1):
        pr = [p.double().requires_grad_(True) for p in params]
        ref = sum(p.abs().sum() for p in pr) if mode == 0 else sum(p.norm(2) for p in pr) / len(pr)
        (ref * 0.37).backward()
        pd = [p.to(DEV).requires_grad_(True) for p in params]
        out = ops.ParamNormFn.apply(mode, *pd)
        (out * 0.37).backward()
        torch.cuda.synchronize()
        assert rel_err(out, ref) < 1e-6
        for a, b in zip(pd, pr):
            assert torch.isfinite(a.grad).all()
            if b.grad.abs().max() == 0:
                assert a.grad.abs().max().item() == 0.0
            else:
                assert rel_err(a.grad, b.grad) < 1e-6


@pytest.mark.parametrize("n,c,hf", [(2, 3, 111), (1, 6, 37)])
def test_convT_out_bwd_data_emits_bn_backward_sums(C, n, c, hf):
    """The data-gradient epilogue's per-tile partials (x_raw given) reduce to the same BatchNorm-backward sums, dgamma and
    dbeta as the stand-alone pass srlz_bn_relu_bwd_sums over (x_raw, dA)."""
    g = torch.Generator().manual_seed(31 + hf)
    himg = (hf - 1) * 2 + 4
    x_raw = (torch.randn(n, hf, hf, 64, generator=g) * 1.3 + 0.2).to(DEV)
    w = (torch.randn(64, c, 4, 4, generator=g) * 0.1).to(DEV)
    dimg = torch.randn(n, c, himg, himg, generator=g).to(DEV)
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    mean, var = x_raw.double().mean((0, 1, 2)).cpu(), x_raw.double().var((0, 1, 2), unbiased=False).cpu()
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    bnp = torch.cat((mean, invstd, gamma.double() * invstd, beta.double() - mean * gamma.double() * invstd)).float().to(DEV)
    d = C.SkinnyDesc(n, c, himg, himg, hf, hf, 1)
    st = C.stream()
    da0, da1 = torch.empty(n, hf, hf, 64, device=DEV), torch.empty(n, hf, hf, 64, device=DEV)
    C.convT_out_bwd_data(C.ptr(dimg), C.ptr(w), C.ptr(da0), None, None, None, d, st)
    partial = torch.full((C.skinny_tiles(d), 128), float("nan"), device=DEV)
    C.convT_out_bwd_data(C.ptr(dimg), C.ptr(w), C.ptr(da1), C.ptr(x_raw), C.ptr(bnp), C.ptr(partial), d, st)
    nb = C.bn_bwd_workspace(0)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    out = [[torch.empty(k, device=DEV) for k in (128, 64, 64)] for _ in range(2)]
    C.bn_relu_bwd_sums(C.ptr(x_raw), C.ptr(bnp), C.ptr(da0), C.ptr(out[0][0]), C.ptr(out[0][1]), C.ptr(out[0][2]), C.ptr(ws), nb,
                       n * hf * hf, 1, st)
    C.bn_bwd_finalize_partials(C.ptr(partial), partial.shape[0], 1, C.ptr(out[1][0]), C.ptr(out[1][1]), C.ptr(out[1][2]), C.ptr(ws),
                               nb, st)
    torch.cuda.synchronize()
    assert torch.equal(da0, da1)
    for a, b in zip(out[0], out[1]):
        assert rel_err(b, a) < 2e-5


@pytest.mark.parametrize("n,c,hf,groups", [(2, 3, 111, 1), (4, 3, 37, 2), (3, 3, 21, 1), (2, 6, 111, 2), (3, 6, 18, 1), (1, 3, 5, 1)])
def test_convT_out_bwd_fused_matches_the_two_launches(C, n, c, hf, groups):
    """srlz_convT_out_bwd_fused (one pass over dy and x_raw) == srlz_convT_out_bwd_data(x_raw, bnp, partial) followed by
    srlz_convT_out_bwd_weight(bnp), for 3 and 6 image channels: dA, BatchNorm-backward sums / dgamma / dbeta (one record per strip
    of the output-stationary kernel), weight and bias gradients to rounding (another contraction / summation order; the bias
    gradient comes from the rows the fused kernel stages: every image pixel owned by exactly one strip), a second launch bit for
    bit, and the fp64 torch oracle."""
    g = torch.Generator().manual_seed(71 + hf)
    himg = (hf - 1) * 2 + 4
    x_raw = (torch.randn(n, hf, hf, 64, generator=g) * 1.3 + 0.2).to(DEV)
    w = (torch.randn(64, c, 4, 4, generator=g) * 0.1).to(DEV)
    dimg = torch.randn(n, c, himg, himg, generator=g).to(DEV)
    recs = []
    for gi in range(groups):
        xg = x_raw[gi * (n // groups):(gi + 1) * (n // groups)].double()
        gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
        mean, var = xg.mean((0, 1, 2)).cpu(), xg.var((0, 1, 2), unbiased=False).cpu()
        invstd = 1.0 / torch.sqrt(var + 1e-5)
        recs.append(torch.cat((mean, invstd, gamma.double() * invstd, beta.double() - mean * gamma.double() * invstd)).float())
    bnp = torch.cat(recs).to(DEV)
    d = C.SkinnyDesc(n, c, himg, himg, hf, hf, 1, groups)
    assert C.convT_out_bwd_fused_supported(d) == 1
    st = C.stream()
    nb = C.bn_bwd_workspace(0)
    wsb = torch.empty(nb, dtype=torch.uint8, device=DEV)
    # reference: the two launches
    da0 = torch.empty(n, hf, hf, 64, device=DEV)
    p0 = torch.empty((C.skinny_tiles(d), 128), device=DEV)
    C.convT_out_bwd_data(C.ptr(dimg), C.ptr(w), C.ptr(da0), C.ptr(x_raw), C.ptr(bnp), C.ptr(p0), d, st)
    n0 = C.skinny_bwd_weight_workspace(d)
    ws0 = torch.empty(n0, dtype=torch.uint8, device=DEV)
    dw0, db0 = torch.empty(64, c, 4, 4, device=DEV), torch.empty(c, device=DEV)
    C.convT_out_bwd_weight(C.ptr(x_raw), C.ptr(dimg), C.ptr(dw0), C.ptr(db0), C.ptr(bnp), C.ptr(ws0), n0, d, st)
    # fused
    da1 = torch.full((n, hf, hf, 64), float("nan"), device=DEV)
    p1 = torch.full((C.convT_out_bwd_fused_tiles(d), 128), float("nan"), device=DEV)
    n1 = C.convT_out_bwd_fused_workspace(d)
    ws1 = torch.empty(n1, dtype=torch.uint8, device=DEV)
    dw1, db1 = torch.full((64, c, 4, 4), float("nan"), device=DEV), torch.full((c,), float("nan"), device=DEV)
    C.convT_out_bwd_fused(C.ptr(dimg), C.ptr(w), C.ptr(da1), C.ptr(x_raw), C.ptr(bnp), C.ptr(p1), C.ptr(dw1), C.ptr(db1), C.ptr(ws1),
                          n1, None, 1.0, 1.0, d, st)
    da2, p2 = torch.full_like(da1, float("nan")), torch.full_like(p1, float("nan"))
    dw2, db2 = torch.full_like(dw1, float("nan")), torch.full_like(db1, float("nan"))
    C.convT_out_bwd_fused(C.ptr(dimg), C.ptr(w), C.ptr(da2), C.ptr(x_raw), C.ptr(bnp), C.ptr(p2), C.ptr(dw2), C.ptr(db2), C.ptr(ws1),
                          n1, None, 1.0, 1.0, d, st)
    out = [[torch.empty(k, device=DEV) for k in (128 * groups, 64, 64)] for _ in range(2)]
    for p, o in ((p0, out[0]), (p1, out[1])):
        C.bn_bwd_finalize_partials(C.ptr(p), p.shape[0], groups, C.ptr(o[0]), C.ptr(o[1]), C.ptr(o[2]), C.ptr(wsb), nb, st)
    torch.cuda.synchronize()
    assert torch.isfinite(da1).all() and torch.isfinite(p1).all() and rel_err(da1, da0) < 5e-6
    assert torch.equal(da2, da1) and torch.equal(p2, p1) and torch.equal(dw2, dw1) and torch.equal(db2, db1)  # deterministic
    for a, b in zip(out[0], out[1]):
        assert rel_err(b, a) < 2e-5
    assert rel_err(dw1, dw0) < 2e-5
    ref = dimg.double().sum((0, 2, 3))
    tol = 3e-6 * (n * himg * himg) ** 0.5  # unit-variance data: a few fp32 ulps of the typical |sum|
    assert float((db0.double() - ref).abs().max()) <= tol
    assert float((db1.double() - ref).abs().max()) <= tol
    # ---- and against the torch oracle (fp64 autograd of ConvTranspose2d(64, 3, 4, 2) on relu(bn(x_raw)), per BatchNorm group): the
    # data gradient w.r.t. the activated input and the weight gradient — not only against the product's own two-launch path
    wr = w.double().cpu().requires_grad_(True)
    per = n // groups
    da_ref = []
    for gi in range(groups):
        sc, sh = recs[gi][128:192].double().view(1, 64, 1, 1), recs[gi][192:].double().view(1, 64, 1, 1)
        a = torch.relu(nchw(x_raw[gi * per:(gi + 1) * per]).double().cpu() * sc + sh).requires_grad_(True)
        F.conv_transpose2d(a, wr, None, stride=2).backward(dimg[gi * per:(gi + 1) * per].double().cpu())
        da_ref.append(a.grad)
    assert rel_err(nchw(da1), torch.cat(da_ref)) < 2e-5
    assert rel_err(dw1, wr.grad) < 2e-5


@pytest.mark.parametrize("n,c,hf,mean,groups", [(2, 3, 111, 1, 2), (4, 3, 37, 0, 2), (2, 6, 21, 1, 1), (6, 3, 111, 1, 2)])
def test_convT_out_forward_with_the_loss_in_its_epilogue(C, n, c, hf, mean, groups):
    """srlz_convT_out_fwd_loss (K11 / K12: the reconstruction / generation loss of the step's two frames taken where the last
    ConvTranspose holds its output) against the un-fused chain srlz_convT_out_fwd -> srlz_sqdiff_pair_loss ->
    srlz_sqdiff_grad_groups, and the torch oracle: the stored error is dec - target bit for bit, the optional reconstruction is
    srlz_convT_out_fwd's bit for bit, the loss agrees to fp32 rounding (fp64 partial sums in a different fixed order), and the
    gradient formed from the error is the un-fused gradient bit for bit (srlz_scale_by_scalar; the fused backward kernel is
    checked in test_convT_out_bwd_fused_with_the_loss_gain)."""
    g = torch.Generator().manual_seed(5 * hf + c)
    himg = (hf - 1) * 2 + 4
    x = (torch.randn(n, hf, hf, 64, generator=g) * 1.1 + 0.1).to(DEV)
    w = (torch.randn(64, c, 4, 4, generator=g) * 0.1).to(DEV)
    b = torch.randn(c, generator=g).to(DEV)
    tgt = (torch.randn(n, c, himg, himg, generator=g) * 1.3 + 0.2).to(DEV)
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    recs = []
    for gi in range(groups):
        xg = x[gi * (n // groups):(gi + 1) * (n // groups)].double().cpu()
        m, v = xg.mean((0, 1, 2)), xg.var((0, 1, 2), unbiased=False)
        inv = 1.0 / torch.sqrt(v + 1e-5)
        recs.append(torch.cat((m, inv, gamma.double() * inv, beta.double() - m * gamma.double() * inv)).float())
    bnp = torch.cat(recs).to(DEV)
    d = C.SkinnyDesc(n, c, himg, himg, hf, hf, 1, groups)
    st = C.stream()
    # ---- un-fused chain
    dec0 = torch.empty(n, c, himg, himg, device=DEV)
    C.convT_out_fwd(C.ptr(x), C.ptr(w), C.ptr(b), C.ptr(dec0), C.ptr(bnp), d, st)
    nb = C.reduce_workspace(dec0.numel())
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    sums0, comb0 = torch.empty(2, device=DEV), torch.empty((), device=DEV)
    per = dec0.numel() // 2
    C.sqdiff_pair_loss(C.ptr(dec0), C.ptr(tgt), per, mean, C.ptr(sums0), C.ptr(comb0), C.ptr(ws), nb, st)
    up = torch.tensor(0.37, device=DEV)
    div = float(per) if mean else 1.0
    grad0 = torch.empty_like(dec0)
    C.sqdiff_grad_groups(C.ptr(dec0), C.ptr(tgt), C.ptr(up), 0, div, 2.0, C.ptr(grad0), per, 2, st)
    # ---- fused
    err = torch.full((n, c, himg, himg), float("nan"), device=DEV)
    dec1 = torch.full((n, c, himg, himg), float("nan"), device=DEV)
    nwg = C.convT_out_fwd_loss_workgroups(d)
    assert nwg > 0
    part = torch.full((2 * nwg,), float("nan"), dtype=torch.float64, device=DEV)
    C.convT_out_fwd_loss(C.ptr(x), C.ptr(w), C.ptr(b), C.ptr(tgt), C.ptr(err), C.ptr(dec1), C.ptr(bnp), C.ptr(part), d, st)
    sums1, comb1 = torch.empty(2, device=DEV), torch.empty((), device=DEV)
    C.pair_loss_finalize(C.ptr(part), nwg, per, mean, C.ptr(sums1), C.ptr(comb1), st)
    grad1 = torch.empty_like(err)
    C.scale_by_scalar(C.ptr(err), C.ptr(up), div, 2.0, C.ptr(grad1), err.numel(), st)
    # the reconstruction may be skipped (the training path): same error, same partials
    err2 = torch.full_like(err, float("nan"))
    part2 = torch.full_like(part, float("nan"))
    C.convT_out_fwd_loss(C.ptr(x), C.ptr(w), C.ptr(b), C.ptr(tgt), C.ptr(err2), None, C.ptr(bnp), C.ptr(part2), d, st)
    torch.cuda.synchronize()
    assert torch.equal(dec1, dec0)
    assert torch.equal(err, dec0 - tgt) and torch.equal(err2, err) and torch.equal(part2, part)
    assert torch.equal(grad1, grad0)
    assert rel_err(sums1, sums0) < 2e-7 and abs(float(comb1) - float(comb0)) <= 2e-7 * abs(float(comb0))
    # oracle: F.conv_transpose2d on relu(bn(x)) in fp64, per-frame sums of squares
    xs = nchw(x).double().cpu()
    half = n // groups
    act = torch.cat([torch.relu(xs[gi * half:(gi + 1) * half] * recs[gi][128:192].double().view(1, 64, 1, 1) +
                                recs[gi][192:].double().view(1, 64, 1, 1)) for gi in range(groups)])
    ref = F.conv_transpose2d(act, w.double().cpu(), b.double().cpu(), stride=2)
    sq = ((ref - tgt.double().cpu()) ** 2).reshape(2, -1).sum(1)
    assert rel_err(sums1, sq) < 1e-5
    expect = (sq[0] / per + sq[1] / per) if mean else (sq[0] + sq[1])
    assert abs(float(comb1) - float(expect)) <= 1e-5 * abs(float(expect))


def test_convT_out_bwd_fused_with_the_loss_gain(C):
    """srlz_convT_out_bwd_fused fed with the stored error + (upstream, div, coef) == the same kernel fed with the materialised
    gradient ((upstream / div) * coef) * error: every output bit for bit."""
    n, hf = 4, 37
    g = torch.Generator().manual_seed(99)
    himg = (hf - 1) * 2 + 4
    x_raw = (torch.randn(n, hf, hf, 64, generator=g) * 1.3 + 0.2).to(DEV)
    w = (torch.randn(64, 3, 4, 4, generator=g) * 0.1).to(DEV)
    err = torch.randn(n, 3, himg, himg, generator=g).to(DEV)
    bnp = torch.cat([torch.cat((torch.randn(64, generator=g) * 0.1, torch.rand(64, generator=g) + 0.5,
                                torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2)) for _ in range(2)]).to(DEV)
    up = torch.tensor(1.7, device=DEV)
    div, coef = float(err.numel() // 2), 2.0
    d = C.SkinnyDesc(n, 3, himg, himg, hf, hf, 1, 2)
    st = C.stream()
    grad = torch.empty_like(err)
    C.scale_by_scalar(C.ptr(err), C.ptr(up), div, coef, C.ptr(grad), err.numel(), st)
    outs = []
    for src, gain in ((grad, (None, 1.0, 1.0)), (err, (C.ptr(up), div, coef))):
        da = torch.full((n, hf, hf, 64), float("nan"), device=DEV)
        p = torch.full((C.convT_out_bwd_fused_tiles(d), 128), float("nan"), device=DEV)
        nws = C.convT_out_bwd_fused_workspace(d)
        ws = torch.empty(nws, dtype=torch.uint8, device=DEV)
        dw, db = torch.full((64, 3, 4, 4), float("nan"), device=DEV), torch.full((3,), float("nan"), device=DEV)
        C.convT_out_bwd_fused(C.ptr(src), C.ptr(w), C.ptr(da), C.ptr(x_raw), C.ptr(bnp), C.ptr(p), C.ptr(dw), C.ptr(db), C.ptr(ws),
                              nws, gain[0], gain[1], gain[2], d, st)
        outs.append((da, p, dw, db))
    torch.cuda.synchronize()
    for a, b in zip(outs[0], outs[1]):
        assert torch.isfinite(a).all() and torch.equal(a, b)


@pytest.mark.parametrize("n,c,h", [(2, 3, 224), (1, 6, 64), (2, 3, 50)])
def test_conv1_bwd_data(C, n, c, h):
    """srlz_conv1_bwd_data: d(loss)/d(image) of Conv2d(C,64,7,2,3) (two-phase pixel-GEMM + gather) against fp64 autograd."""
    g = torch.Generator().manual_seed(900 + h + c)
    x = torch.randn(n, c, h, h, generator=g)
    w = torch.randn(64, c, 7, 7, generator=g) * 0.1
    hf = (h + 6 - 7) // 2 + 1
    dy = torch.randn(n, 64, hf, hf, generator=g)
    xr = x.double().requires_grad_(True)
    F.conv2d(xr, w.double(), None, stride=2, padding=3).backward(dy.double())
    d = C.SkinnyDesc(n, c, h, h, hf, hf, 0)
    dx = torch.full((n, c, h, h), float("nan"), device=DEV)
    wd, dyd = w.to(DEV), nhwc(dy).to(DEV)
    C.conv1_bwd_data(C.ptr(dyd), C.ptr(wd), C.ptr(dx), d, C.stream())
    torch.cuda.synchronize()
    assert rel_err(dx, xr.grad) < 2e-5


def test_native_comm_single_rank(C):
    """srlz_comm_* over RCCL (include/srlz.h) with a one-rank communicator on this GPU: unique id, init, in-place sum
    all-reduce (the identity for one rank), world size, destroy — the C-ABI path a host without torch.distributed binds
    (SRLZ_COMM=rccl routes the training step's bucket through it; >= 2 ranks are covered by tests/test_ddp_gpu.py)."""
    import ctypes
    assert C.comm_world() == 0
    n = C.comm_unique_id_bytes()
    assert n == 128
    buf = ctypes.create_string_buffer(n)
    C.comm_unique_id(buf)
    assert any(b != 0 for b in buf.raw)
    C.comm_init(buf, 0, 1)
    try:
        assert C.comm_world() == 1
        with pytest.raises(C.SrlzError):
            C.comm_init(buf, 0, 1)  # one communicator per process
        x = torch.randn(1 << 20, device=DEV)
        ref = x.clone()
        C.comm_allreduce_f32(C.ptr(x), x.numel(), C.stream())
        torch.cuda.synchronize()
        assert torch.equal(x, ref)
    finally:
        C.comm_destroy()
    assert C.comm_world() == 0


def test_total_loss_is_pythons_left_to_right_fp32_sum(C):
    """ops.TotalLossFn == LossManager.computeTotalLoss()'s expression sum([w_i * l_i]) on 0-dim fp32 tensors, bit for bit (value,
    scalar tail, gradients)."""
    from srlz import ops
    g = torch.Generator().manual_seed(5)
    vals = [(torch.randn((), generator=g) * s).to(DEV).requires_grad_() for s in (5.0, 3e6, 1e3, 0.7, 1e-3)]
    weights = (1.0, 0.5e-6, 2.0, 1.0, 0.3)
    ref_terms = [v.detach().clone().requires_grad_() for v in vals]
    ref = sum([weights[i] * ref_terms[i] for i in range(len(weights))])
    (ref * 1.5).backward()
    tail = torch.full((16,), float("nan"), device=DEV)
    tot = ops.TotalLossFn.apply(weights, tail, *vals)
    (tot * 1.5).backward()
    torch.cuda.synchronize()
    assert torch.equal(tot.detach(), ref.detach())
    assert torch.equal(tail[0], ref.detach()) and all(torch.equal(tail[1 + i], vals[i].detach()) for i in range(len(vals)))
    for a, b in zip(vals, ref_terms):
        assert torch.equal(a.grad, b.grad)



@pytest.mark.parametrize("n,hi,groups,training,zero_gamma", [(2, 27, 1, 1, False), (8, 27, 2, 1, True), (6, 55, 2, 1, False),
                                                           (32, 13, 2, 1, False), (4, 27, 2, 0, True)])
def test_conv64_bwd_fused_whole_block_backward(C, n, hi, groups, training, zero_gamma):
    """srlz_conv64_bwd_fused — data, weight and bias gradient of ConvTranspose2d(64,64,3,2) whose output went through
    BatchNorm2d + ReLU and whose input was relu(batchnorm(x)), from ONE staging of the rebuilt d(loss)/dy — against
    (i) fp64 autograd through relu(bn(x)) -> conv_transpose2d -> batch_norm -> relu per BatchNorm group, EVERY element of dx
    and dw to 1e-4 at every size, and
    (ii) the two-launch path srlz_conv64_bwd_data(dy_out) + srlz_conv64_bwd_weight: dx bit for bit, dw / db to rounding.

    The discrete decisions are taken out of the comparison the way tests/test_step_gpu.py does it for whole steps (two correct
    evaluations decide a ReLU at |bn(.)| ~ 1e-7 differently, and one flipped decision moves ~256 dx values by 1e-2 of the
    maximum): the INPUT activations are moved off the ReLU threshold of their record (the record is an input of the kernel), and
    the incoming gradient dA is zeroed wherever the device's own bn(y) lies within 1e-4 of the output ReLU's threshold — there the
    decision multiplies a zero in the data gradient, in the weight gradient and in both BatchNorm-backward sums, for either
    evaluation; everywhere else the fp32 and fp64 decisions agree (the two evaluations of bn(y) differ by ~1e-6)."""
    assert n % groups == 0
    g = torch.Generator().manual_seed(77 * hi + n)
    ho = (hi - 1) * 2 + 3
    x = torch.randn(n, 64, hi, hi, generator=g) * 1.2 + 0.1            # raw output of the previous layer
    w, b = torch.randn(64, 64, 3, 3, generator=g) * 0.05, torch.randn(64, generator=g) * 0.1
    gx, bx = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    if zero_gamma:  # input-side BatchNorm channels without a usable scale: exactly 0 (with a positive and a negative shift) and tiny
        gx[3], bx[3] = 0.0, 0.3
        gx[17], bx[17] = 0.0, -0.2
        gx[40], bx[40] = 1e-5, 0.25
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2
    rm, rv = torch.randn(64, generator=g) * 0.1, torch.rand(64, generator=g) + 0.5
    da = torch.randn(n, 64, ho, ho, generator=g)
    per = n // groups
    # ---- the input's BatchNorm record per group (fp32, as the kernel receives it); x moved off its ReLU threshold
    xrecs, recs64 = [], []
    for gi in range(groups):
        xs = x[gi * per:(gi + 1) * per].double()
        m, v = xs.mean((0, 2, 3)), xs.var((0, 2, 3), unbiased=False)
        inv = 1.0 / torch.sqrt(v + 1e-5)
        rec = torch.cat((m, inv, gx.double() * inv, bx.double() - m * gx.double() * inv)).float()
        xrecs.append(rec)
        sc, sh = rec[128:192].double().view(1, 64, 1, 1), rec[192:256].double().view(1, 64, 1, 1)
        recs64.append((sc, sh))
        z = xs * sc + sh
        near = z.abs() < 1e-3
        movable = (sc.abs() > 1e-2).expand_as(z)  # (a channel without a scale cannot be moved off the threshold — nor does it sit on it)
        xs = torch.where(near & movable, (torch.where(z >= 0, 2e-3, -2e-3) - sh) / torch.where(sc.abs() > 1e-2, sc, torch.ones(())), xs)
        x[gi * per:(gi + 1) * per] = xs.float()
        assert float(((x[gi * per:(gi + 1) * per].double() * sc + sh).abs()).min()) > 5e-4
    # ---- device: forward through the C ABI to get y, its statistics and the backward sums
    st = C.stream()
    d = C.Conv64Desc(n, hi, hi, ho, ho, 3, 2, 0, 1, groups)
    assert C.conv64_bwd_fused_supported(d) == 1
    xd, wd, bd = nhwc(x).to(DEV), w.to(DEV), b.to(DEV)
    xbnp = torch.cat(xrecs).to(DEV)
    packs = torch.empty(2, C.conv64_packed_floats(), device=DEV)
    C.conv64_pack_weights(C.ptr(wd), C.ptr(packs[0]), C.ptr(packs[1]), d, st)
    y = torch.empty(n, ho, ho, 64, device=DEV)
    stats = torch.empty(C.conv64_fwd_tiles(d), 128, device=DEV)
    C.conv64_fwd(C.ptr(xd), C.ptr(packs[0]), C.ptr(bd), C.ptr(y), C.ptr(stats), C.ptr(xbnp), d, st)
    gd, bed, rmd, rvd = gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV)
    bnp = torch.empty(256 * groups, device=DEV)
    nbn = C.bn_bwd_workspace(0)
    bws = torch.empty(nbn, dtype=torch.uint8, device=DEV)
    if training:
        bstat = torch.empty(128 * groups, device=DEV)
        C.bn_finalize(C.ptr(stats), stats.shape[0], groups, per * ho * ho, C.ptr(gd), C.ptr(bed), 1e-5, 0.1, 1, C.ptr(rmd), C.ptr(rvd),
                      None, C.ptr(bnp), C.ptr(bstat), C.ptr(bws), nbn, st)
    else:
        one = torch.empty(256, device=DEV)
        C.bn_eval_params(C.ptr(gd), C.ptr(bed), C.ptr(rmd), C.ptr(rvd), 1e-5, C.ptr(one), st)
        bnp = one.repeat(groups)
    # ---- dA = 0 wherever the device's bn(y) is within 1e-4 of the output ReLU's threshold
    torch.cuda.synchronize()
    rec_y = bnp.view(groups, 256).double().cpu()
    zy = nchw(y).double().cpu().view(groups, per, 64, ho, ho) * rec_y[:, 128:192].view(groups, 1, 64, 1, 1) \
        + rec_y[:, 192:256].view(groups, 1, 64, 1, 1)
    tie = (zy.abs() < 1e-4).view(n, 64, ho, ho)
    da = torch.where(tie, torch.zeros(()), da)
    dad = nhwc(da).to(DEV)
    # ---- fp64 reference, group by group (per-call BatchNorm statistics)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    dx_ref = []
    for gi in range(groups):
        sc, sh = recs64[gi]
        a = torch.relu(x[gi * per:(gi + 1) * per].double() * sc + sh).requires_grad_(True)
        yr = F.conv_transpose2d(a, wr, br, stride=2)
        zr = F.batch_norm(yr, rm.double().clone(), rv.double().clone(), gamma.double(), beta.double(), bool(training), 0.1, 1e-5)
        # outside the zeroed ties both evaluations take the same decision
        assert bool((((zr > 0) == (zy[gi] > 0)) | tie[gi * per:(gi + 1) * per]).all())
        F.relu(zr).backward(da[gi * per:(gi + 1) * per].double())
        dx_ref.append(a.grad)
    dx_ref = torch.cat(dx_ref)
    sums, dgm, dbt = torch.empty(128 * groups, device=DEV), torch.empty(64, device=DEV), torch.empty(64, device=DEV)
    C.bn_relu_bwd_sums(C.ptr(y), C.ptr(bnp), C.ptr(dad), C.ptr(sums), C.ptr(dgm), C.ptr(dbt), C.ptr(bws), nbn, n * ho * ho, groups, st)
    # ---- two launches
    dy_out = torch.full((n, ho, ho, 64), float("nan"), device=DEV)
    op = C.BnBwdOperand(y.data_ptr(), bnp.data_ptr(), sums.data_ptr(), per * ho * ho, training, dy_out.data_ptr())
    dx0 = torch.full((n, hi, hi, 64), float("nan"), device=DEV)
    C.conv64_bwd_data(C.ptr(dad), C.ptr(packs[1]), C.ptr(dx0), op, d, st)
    nbytes = C.conv64_bwd_weight_workspace(d)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    dw0, db0 = torch.full((64, 64, 3, 3), float("nan"), device=DEV), torch.full((64,), float("nan"), device=DEV)
    C.conv64_bwd_weight(C.ptr(xd), C.ptr(dy_out), C.ptr(dw0), C.ptr(db0), C.ptr(xbnp), None, C.ptr(ws), nbytes, d, st)
    # ---- one launch
    op1 = C.BnBwdOperand(y.data_ptr(), bnp.data_ptr(), sums.data_ptr(), per * ho * ho, training, None)
    dx1 = torch.full((n, hi, hi, 64), float("nan"), device=DEV)
    dw1, db1 = torch.full((64, 64, 3, 3), float("nan"), device=DEV), torch.full((64,), float("nan"), device=DEV)
    nb1 = C.conv64_bwd_fused_workspace(d)
    ws1 = torch.full((nb1 // 4,), float("nan"), device=DEV)
    rows = C.conv64_bwd_fused_bn_rows(d)
    part1 = torch.full((rows, 128), float("nan"), device=DEV)
    C.conv64_bwd_fused(C.ptr(xd), C.ptr(xbnp), C.ptr(dad), op1, C.ptr(packs[1]), C.ptr(dx1), C.ptr(dw1), C.ptr(db1), C.ptr(part1), C.ptr(ws1), nb1,
                       d, st)
    # determinism: a second launch, bit for bit (this one without the BatchNorm-backward records: the other outputs do not depend on them)
    dx2, dw2, db2 = torch.empty_like(dx1), torch.empty_like(dw1), torch.empty_like(db1)
    C.conv64_bwd_fused(C.ptr(xd), C.ptr(xbnp), C.ptr(dad), op1, C.ptr(packs[1]), C.ptr(dx2), C.ptr(dw2), C.ptr(db2), None, C.ptr(ws1), nb1, d, st)
    torch.cuda.synchronize()
    assert torch.isfinite(dx1).all() and torch.equal(dx1, dx0)
    assert torch.equal(dx2, dx1) and torch.equal(dw2, dw1) and torch.equal(db2, db1)
    # ---- round 6: the BatchNorm-backward sums of the layer that produced x, out of the same launch's flush (+ its companion for the
    # channels without a usable scale), against srlz_bn_relu_bwd_sums' separate pass over (x, dx) — sums per group, dgamma / dbeta
    assert torch.isfinite(part1).all()
    sums_a, dg_a, db_a = torch.empty(128 * groups, device=DEV), torch.empty(64, device=DEV), torch.empty(64, device=DEV)
    C.bn_bwd_finalize_partials(C.ptr(part1), rows, groups, C.ptr(sums_a), C.ptr(dg_a), C.ptr(db_a), C.ptr(bws), nbn, st)
    sums_b, dg_b, db_b = torch.empty(128 * groups, device=DEV), torch.empty(64, device=DEV), torch.empty(64, device=DEV)
    C.bn_relu_bwd_sums(C.ptr(xd), C.ptr(xbnp), C.ptr(dx1), C.ptr(sums_b), C.ptr(dg_b), C.ptr(db_b), C.ptr(bws), nbn, n * hi * hi, groups, st)
    torch.cuda.synchronize()
    for got, want in ((sums_a, sums_b), (dg_a, dg_b), (db_a, db_b)):
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 2e-5 * scale, float((got - want).abs().max()) / scale
    if zero_gamma:  # those channels DID go through the companion (the activation is the constant max(shift, 0) there)
        assert float(sums_b.view(groups, 128)[:, 3].abs().min()) > 0 and float(sums_b.view(groups, 128)[:, 17].abs().max()) == 0.0
    # fp64 oracle with the decisions out of the way (docstring): EVERY element, every size
    dev_dx = (nchw(dx1).double().cpu() - dx_ref).abs() / float(dx_ref.abs().max())
    assert float(dev_dx.max()) < 1e-4, float(dev_dx.max())
    assert rel_err(dw1, dw0) < 2e-5
    assert rel_err(dw1, wr.grad) < 1e-4, rel_err(dw1, wr.grad)
    scale = float(da.abs().sum()) / 64
    assert float((db1 - db0).abs().max()) < 1e-5 * scale
    if training:  # the bias gradient of a convolution followed by train-mode BatchNorm is identically zero
        assert float(db1.abs().max()) < 1e-3 * scale
    else:
        assert rel_err(db1, br.grad) < 5e-5


@pytest.mark.parametrize("n,h,pool_pad,stride,groups,training,zero_gamma", [
    (4, 112, 1, 1, 2, True, False),   # block 1 -> conv2 (3x3 s1 p1 on the 56x56 pooled map), the pair's two BatchNorm groups
    (2, 56, 0, 2, 1, True, False),    # block 2 -> conv3 (3x3 s2 p1 on the 27x27 pooled map): a four-class scatter data gradient
    (6, 56, 0, 2, 2, True, True),     # a channel whose BatchNorm scale is exactly 0 (xhat from the convolution output under the argmax)
    (4, 112, 1, 1, 2, True, "tiny"),  # gamma ~ 1e-6 against beta ~ 0.3: (z - shift) / scale would cancel (advisor, round 3)
    (2, 40, 1, 1, 1, False, False),   # eval mode (validation minibatches backpropagate too)
])
def test_pool_block_bn_backward_sums_from_the_next_convs_data_gradient(C, n, h, pool_pad, stride, groups, training, zero_gamma):
    """ops.PoolLink: BatchNorm -> ReLU -> MaxPool(3, 2) followed by conv3x3 — the block's two BatchNorm-backward sums taken in the
    epilogue of the convolution's data-gradient launch (srlz_conv64_bwd_data_pool_sums + srlz_bn_bwd_finalize_partials) against
    (i) the separate pass (srlz_bn_relu_pool_bwd: same product path with the link off): dy of the pooled block, dgamma, dbeta,
    and bit-identical d(pooled); (ii) fp64 autograd of the same chain."""
    from srlz import ops
    g = torch.Generator().manual_seed(31 * h + n)
    y0 = torch.randn(n, 64, h, h, generator=g) * 1.3 + 0.2
    gamma0, beta0 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3
    if zero_gamma == "tiny":
        gamma0[7], beta0[7] = 1e-6, 0.3      # relu(bn(.)) = 0.3 +- 1e-6 * xhat: every pooled value positive, xhat lost in its rounding
        gamma0[21], beta0[21] = -3e-6, 0.2
        gamma0[50], beta0[50] = 2e-6, -0.1   # nothing passes the ReLU
    elif zero_gamma:
        gamma0[5] = 0.0
        beta0[5] = 0.25   # relu(bn(.)) = 0.25 everywhere in that channel: every window's first position is the argmax
        gamma0[40] = 0.0
        beta0[40] = -0.1  # ... and here nothing passes the ReLU
    w0 = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    hp = (h + 2 * pool_pad - 3) // 2 + 1
    ho = (hp + 2 - 3) // stride + 1
    dz0 = torch.randn(n, 64, ho, ho, generator=g)

    # fp64 reference, one BatchNorm call per group
    yr, gr, br, wr = (t.double().requires_grad_(True) for t in (y0, gamma0, beta0, w0))
    outs = []
    per = n // groups
    for gi in range(groups):
        ys = yr[gi * per:(gi + 1) * per]
        z = F.batch_norm(ys, torch.zeros(64, dtype=torch.float64), torch.ones(64, dtype=torch.float64), gr, br, training, 0.1, 1e-5)
        outs.append(F.conv2d(F.max_pool2d(F.relu(z), 3, 2, pool_pad), wr, None, stride=stride, padding=1))
    (torch.cat(outs) * dz0.double()).sum().backward()

    def run(link_on):
        yd = nhwc(y0).to(DEV).requires_grad_(True)
        gd, bd, wd = (t.clone().to(DEV).requires_grad_(True) for t in (gamma0, beta0, w0))
        rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
        # per-tile statistics of y as the producing convolution would have left them (training mode)
        with ops.batch_groups(groups):
            st = None
            if training:
                d1 = ops.conv64_desc(n, h, h, 1, 1, False, groups)
                ntiles = C.conv64_fwd_tiles(d1)
                st = torch.zeros((ntiles, 128), dtype=torch.float32, device=DEV)
                tpg = ntiles // groups
                for gi in range(groups):
                    ysl = yd.detach()[gi * per:(gi + 1) * per].reshape(-1, 64).double()
                    st[gi * tpg, :64] = ysl.sum(0).float()
                    st[gi * tpg, 64:] = (ysl * ysl).sum(0).float()
            link = ops.PoolLink() if link_on else None
            p = ops.BNReLUPoolFn.apply(yd, st, gd, bd, rm, rv, training, pool_pad, False, None, link)
            out, _ = ops.Conv64Fn.apply(p, wd, None, stride, 1, False, training, link)  # (want_stats = training, as in hotpath)
            p.retain_grad()
            (out * nhwc(dz0).to(DEV)).sum().backward()
        torch.cuda.synchronize()
        assert (link is None) or (link.record is None and link.partials is None)  # consumed
        return yd.grad, gd.grad, bd.grad, wd.grad, p.grad

    a, b = run(True), run(False)
    assert torch.equal(a[4], b[4]) and torch.equal(a[3], b[3])       # d(pooled), dW: the same launches' arithmetic
    for i, tol in ((0, 2e-5), (1, 2e-5), (2, 2e-5)):                  # dy of the block, dgamma, dbeta: summation order only
        assert rel_err(a[i], b[i]) < tol, (i, rel_err(a[i], b[i]))
    keep = torch.ones(64, dtype=torch.bool)
    if zero_gamma == "tiny":
        # relu(bn(.)) = beta +- 1e-6 * xhat there: fp32 and fp64 break the max-pool's near-ties differently, so autograd routes a few
        # per cent of that channel's gradient elsewhere; those channels are held by (i) above, whose separate pass takes xhat from y
        keep[[7, 21, 50]] = False
    assert rel_err(nchw(a[0]), yr.grad) < 1e-4
    assert rel_err(a[1].cpu()[keep], gr.grad[keep]) < 1e-4 and rel_err(a[2].cpu()[keep], br.grad[keep]) < 1e-4
