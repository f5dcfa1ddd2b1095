"""Autograd seam between the host-side model classes and the HIP kernels (include/srlz.h).

The reference relies on ``loss.backward()`` (models/learner.py:489); here each fused block is one
``torch.autograd.Function`` whose forward/backward enqueue C-ABI calls on the current HIP stream.  PyTorch tensors are
used for storage only: no torch compute op touches an activation.  Activations are NHWC between the first conv and the
last transposed conv; images and everything the caller sees stay in the reference's NCHW layout.
"""
import weakref

import torch
from torch.autograd import Function

from . import _cabi as C
from ._cabi import ptr, stream, Conv64Desc, SkinnyDesc, PoolDesc, ConvNDesc

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# ---- BatchNorm groups -------------------------------------------------------------------------------------------------
# The reference calls the model twice per step, self.model(obs) and self.model(next_obs) (models/learner.py:392-393): two
# independent BatchNorm calls.  Inside `with batch_groups(2):` a forward over the CONCATENATED batch [obs ; next_obs] is
# that pair of calls as ONE launch per layer: every kernel treats images [g*N/G, (g+1)*N/G) as group g with its own batch
# statistics / BatchNorm record, running statistics take the groups' momentum updates in order, num_batches_tracked
# advances by G, weight gradients are summed over the groups by the kernels.  Eval mode has no batch statistics: one group.
_GROUPS = 1


class batch_groups(object):
    def __init__(self, groups):
        self.groups = int(groups)

    def __enter__(self):
        global _GROUPS
        self.prev, _GROUPS = _GROUPS, self.groups
        return self

    def __exit__(self, *exc):
        global _GROUPS
        _GROUPS = self.prev
        return False


def cur_groups(training):
    return _GROUPS if training else 1

_workspaces = {}

# ---- optional per-launch timing (bench.py's roofline leg): HIP events recorded on the stream the kernels run on ----
_timers_on = False
_timer_events = []  # (kernel symbol, layer key, algorithmic flop, start event, end event)


def timers_enable(flag):
    global _timers_on
    _timers_on = bool(flag)
    if flag:
        del _timer_events[:]


def _launch(kernel, key, flop, fn):
    if not _timers_on:
        return fn()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    out = fn()
    end.record()
    _timer_events.append((kernel, key, flop, start, end))
    return out


def timers_report():
    """{kernel or kernel/layer: {"ms", "launches", "flop"}} over everything recorded since timers_enable(True)."""
    torch.cuda.synchronize()
    rep = {}
    for kernel, key, flop, start, end in _timer_events:
        ms = start.elapsed_time(end)
        for name in (kernel, "%s/%s" % (kernel, key)):
            r = rep.setdefault(name, {"ms": 0.0, "launches": 0, "flop": 0.0})
            r["ms"] += ms
            r["launches"] += 1
            r["flop"] += flop
    return rep


# Direct gradient delivery (srlz.optim.FlatParams.grad_buffer / deliver): a parameter re-homed into the flat bucket
# carries `_srlz_flat`; its weight-gradient kernel writes straight into a staging copy of the bucket and autograd gets
# None, which removes one tiny `grad += new` launch per parameter and contribution.  Parameters that are not re-homed (a bare module
# without FlatParams) or whose stages are used up (a fourth contribution in one pass) take autograd's classic accumulation; tests
# clear _DIRECT_GRADS in-process to compare the two.
_DIRECT_GRADS = True


def _gbuf(param, shape=None, device=None):
    """Output buffer for the gradient of `param` (a staging view when the parameter lives in a FlatParams bucket)."""
    if param is not None and _DIRECT_GRADS:
        home = getattr(param, "_srlz_flat", None)
        if home is not None:
            buf = home[0].grad_buffer(home[1])
            if buf is not None:
                buf._srlz_staged = True
                return buf
    if param is not None:
        return torch.empty_like(param)
    return torch.empty(shape, dtype=torch.float32, device=device)


def _give(param, grad):
    """What a backward function returns to autograd for `param`: None when `grad` was written into a staging bucket."""
    if grad is not None and getattr(grad, "_srlz_staged", False):
        return None
    return grad


def _conv64_flop(d):
    # algorithmic FLOP (2 x MAC) of the layer: 9 taps x 64 x 64 per position of the low-resolution side
    pos = d.n * (d.hi * d.wi if d.transposed else d.ho * d.wo)
    return 2.0 * 9 * 64 * 64 * pos


def _convT_out_flop(d):
    # ConvTranspose2d(64, C, 4, 2): 16 taps x 64 x C per feature position
    return 2.0 * 16 * 64 * d.c * d.n * d.hf * d.wf


def _conv1_flop(d):
    # Conv2d(C, 64, 7, 2, 3): 49 taps x C x 64 per feature position
    return 2.0 * 49 * 64 * d.c * d.n * d.hf * d.wf


def _skinny_key(d, what):
    return "%s n%d c%d %dx%d<->%dx%d %s" % ("conv1" if d.kind == 0 else "convT5", d.n, d.c, d.himg, d.wimg, d.hf, d.wf, what)


def _conv64_key(d, what):
    return "%s s%d n%d %dx%d->%dx%d %s" % ("convT" if d.transposed else "conv", d.stride, d.n, d.hi, d.wi, d.ho, d.wo, what)


def _ws(nbytes, device, slot=0):
    """Grow-only scratch buffer per (device, current stream, slot): work on one stream is ordered, so a buffer is only
    ever shared by launches of the stream that owns it."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, slot)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


import os as _os


def _check(t, name):
    if t.device.type != "cuda":
        raise C.SrlzError("%s must live on the GPU: the srl-zoo_amd hot path has no CPU fallback" % name)
    if t.dtype != torch.float32:
        raise C.SrlzError("%s must be float32 (got %s)" % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------------------------------------------
# uint8 frames: observations that are still the loader's bytes, [N,C,W,H] planar (the reference's tensor layout,
# preprocessing/data_loader.py:255, before preprocessInput).  conv1 (forward, fused weight gradient) and the fused reconstruction
# loss normalise them through a 3 x 256 table while they stage their windows; everything else asks for the float tensor.
# ----------------------------------------------------------------------------------------------------------------
_NORM_LUT = {}


def norm_lut(device):
    """((v / 255) - mean[c]) / std[c] for v = 0..255, c = R, G, B (preprocessing/utils.py:20-32), one table per device.

    Built ONCE and made visible to every stream before it is returned (the fill is followed by a device synchronisation), so a
    later launch on another stream reads a finished table.  The first request must not fall inside a hipGraph capture — the fill
    would be recorded instead of executed; SRL4robotics.__init__ asks for the table eagerly for that reason."""
    index = device.index if device.index is not None else torch.cuda.current_device()
    lut = _NORM_LUT.get(index)
    if lut is None:
        if torch.cuda.is_current_stream_capturing():
            raise C.SrlzError("norm_lut: first use inside a stream capture; call srlz.ops.norm_lut(device) once before capturing")
        lut = torch.empty((3, 256), dtype=torch.float32, device=torch.device("cuda", index))
        C.normalize_lut(ptr(lut), stream())
        torch.cuda.synchronize(index)
        _NORM_LUT[index] = lut
    return lut


def is_u8_frames(t):
    return isinstance(t, torch.Tensor) and t.dtype == torch.uint8


def _check_u8(t, name):
    if t.device.type != "cuda":
        raise C.SrlzError("%s must live on the GPU: the srl-zoo_amd hot path has no CPU fallback" % name)
    if t.dim() != 4 or t.shape[1] not in (3, 6, 9):
        raise C.SrlzError("%s: uint8 frames must be [N,C,W,H] planar with C in (3, 6, 9), got %s" % (name, tuple(t.shape)))
    return t if t.is_contiguous() else t.contiguous()


def frames_as_float(frames, out=None):
    """uint8 [N,C,W,H] planar frames -> the normalised fp32 observation tensor (float tensors pass through)."""
    if not is_u8_frames(frames):
        return frames
    frames = _check_u8(frames, "frames")
    n, c = frames.shape[:2]
    if out is None:
        out = torch.empty(frames.shape, dtype=torch.float32, device=frames.device)
    elif tuple(out.shape) != tuple(frames.shape) or out.dtype != torch.float32 or not out.is_contiguous():
        raise C.SrlzError("frames_as_float: `out` must be a contiguous float32 tensor of the frames' shape")
    lut, plane = norm_lut(frames.device), frames[0, 0].numel()
    per = max(1, 65535 // c)  # one launch takes n * c <= 65535 planes (grid.y)
    for i in range(0, n, per):
        m = min(per, n - i)
        C.normalize_u8_planar(ptr(frames[i:i + m]), ptr(lut), ptr(out[i:i + m]), m, c, plane, stream())
    return out


# ----------------------------------------------------------------------------------------------------------------
# conv1: nn.Conv2d(C, 64, 7, stride 2, pad 3, bias=False) — models/models.py:49
# ----------------------------------------------------------------------------------------------------------------
def _skinny_desc(n, c, himg, wimg, kind, groups=1):
    if kind == 0:
        hf, wf = (himg + 6 - 7) // 2 + 1, (wimg + 6 - 7) // 2 + 1
    else:
        hf, wf = (himg - 4) // 2 + 1, (wimg - 4) // 2 + 1
    return SkinnyDesc(n, c, himg, wimg, hf, wf, kind, groups)


class Conv1Fn(Function):
    @staticmethod
    def forward(ctx, x, w, want_stats):
        x, w = _check(x, "conv1 input"), _check(w, "conv1 weight")
        n, c, h, wd = x.shape
        d = _skinny_desc(n, c, h, wd, 0, cur_groups(want_stats))
        y = torch.empty((n, d.hf, d.wf, 64), dtype=torch.float32, device=x.device)
        stats = torch.empty((C.skinny_tiles(d), 128), dtype=torch.float32, device=x.device) if want_stats else None
        C.conv1_fwd(ptr(x), ptr(w), ptr(y), ptr(stats), d, stream())
        ctx.save_for_backward(x, w)
        ctx.desc = d
        if stats is None:
            stats = torch.empty(0, device=x.device)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)  # no zero tensor for the statistics output in backward
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        x, w = ctx.saved_tensors
        d = ctx.desc
        dy = _check(dy, "conv1 dy")
        dw = None
        if ctx.needs_input_grad[1]:
            dw = _gbuf(w)
            nbytes = C.skinny_bwd_weight_workspace(d)
            ws = _ws(nbytes, x.device)
            C.conv1_bwd_weight(ptr(x), ptr(dy), ptr(dw), ptr(ws), nbytes, d, stream())
        dx = None
        if ctx.needs_input_grad[0]:  # only when the image carries a gradient (frozen denoiser of the perceptual loss)
            dx = torch.empty_like(x)
            C.conv1_bwd_data(ptr(dy), ptr(w), ptr(dx), d, stream())
        return dx, _give(w, dw), None


# ----------------------------------------------------------------------------------------------------------------
# 64 -> 64 convs: conv3x3 (models/models.py:54,59,217-226) and ConvTranspose2d(64,64,3,2) (models.py:66,70,74,78)
# ----------------------------------------------------------------------------------------------------------------
def conv64_desc(n, hi, wi, stride, pad, transposed, groups=1):
    if transposed:
        ho, wo = (hi - 1) * stride - 2 * pad + 3, (wi - 1) * stride - 2 * pad + 3
    else:
        ho, wo = (hi + 2 * pad - 3) // stride + 1, (wi + 2 * pad - 3) // stride + 1
    return Conv64Desc(n, hi, wi, ho, wo, 3, stride, pad, 1 if transposed else 0, groups)


class PoolLink:
    """Hand-over between an encoder block (conv -> BatchNorm -> ReLU -> MaxPool, producer of a pooled map) and the convolution that
    consumes the pooled map: the consumer's data-gradient launch computes d(pooled) tile by tile and can take the two
    BatchNorm-backward sums of the producing block in its epilogue (srlz_conv64_bwd_data_pool_sums) instead of the producer running a
    pass of its own over (d pooled, pooled, argmax).  The producer's forward leaves (pooled, bnp, y, argmax, pool descriptor) here; the
    consumer's backward leaves the per-tile records; the producer's backward finalises them.  One producer, one consumer, one pass."""

    def __init__(self):
        self.record = None    # (pooled, bnp, y, argmax, pool desc), set by the producer's forward
        self.partials = None  # [tiles, 128] per-tile sums, set by the consumer's backward

    def take_partials(self):
        p, self.partials, self.record = self.partials, None, None
        return p


_POOL_LINK = True  # (the route without it — hotpath.TAPS on, link = None — runs srlz_bn_relu_pool_bwd_sums in the producer instead)


class Conv64Fn(Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, transposed, want_stats, in_link=None):
        x, w = _check(x, "conv64 input"), _check(w, "conv64 weight")
        n, hi, wi, ch = x.shape
        assert ch == 64 and tuple(w.shape) == (64, 64, 3, 3)
        d = conv64_desc(n, hi, wi, stride, pad, transposed, cur_groups(want_stats))
        y = torch.empty((n, d.ho, d.wo, 64), dtype=torch.float32, device=x.device)
        ctx.wino = bool(C.conv64_wino_supported(d))
        if ctx.wino:
            # conv3x3 stride 1 pad 1 on an even map (conv2, models/models.py:54): Winograd F(2x2, 3x3) — 16 multiplications per 2x2 output
            # patch and (ci, co) instead of 36 (csrc/wino.hip), forward and data gradient; the weight gradient stays the direct contraction.
            packs = torch.empty((2, C.conv64_wino_packed_floats()), dtype=torch.float32, device=x.device)  # (G g G^T, both directions)
            C.conv64_wino_pack_weights(ptr(w), ptr(packs[0]), ptr(packs[1]), stream())
            stats = torch.empty((C.conv64_wino_tiles(d), 128), dtype=torch.float32, device=x.device) if want_stats else None
            _launch("conv64_wino_kernel", _conv64_key(d, "fwd"), _conv64_flop(d),
                    lambda: C.conv64_wino_fwd(ptr(x), ptr(packs[0]), ptr(bias), ptr(y), ptr(stats), None, d, stream()))
        else:
            packs = torch.empty((2, C.conv64_packed_floats()), dtype=torch.float32, device=x.device)
            C.conv64_pack_weights(ptr(w), ptr(packs[0]), ptr(packs[1]), d, stream())
            stats = torch.empty((C.conv64_fwd_tiles(d), 128), dtype=torch.float32, device=x.device) if want_stats else None
            name = "conv64_gather_pipe_kernel" if (bias is None and C.conv64_gather_pipe_supported(d, 0)) else "conv64_fwd_kernel"
            _launch(name, _conv64_key(d, "fwd"), _conv64_flop(d),
                    lambda: C.conv64_fwd(ptr(x), ptr(packs[0]), ptr(bias), ptr(y), ptr(stats), None, d, stream()))
        ctx.save_for_backward(x, packs)
        ctx.desc = d
        ctx.has_bias = bias is not None
        ctx.needs_dx = ctx.needs_input_grad[0]
        ctx.params = (w, bias)
        ctx.in_link = in_link  # (x is the pooled map of the block that filled this link)
        if stats is None:
            stats = torch.empty(0, device=x.device)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)  # no zero tensor for the statistics output in backward
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        x, packs = ctx.saved_tensors
        d = ctx.desc
        dy = _check(dy, "conv64 dy")
        dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):  # (a frozen layer needs neither)
            dw = _gbuf(ctx.params[0])
            db = _gbuf(ctx.params[1]) if ctx.has_bias else None
            nbytes = C.conv64_wino_bwd_weight_workspace(d) if (ctx.wino and not ctx.has_bias) else 0
            if nbytes > 0:  # (0: not a shape the transposed Winograd kernel takes — it indexes ALL groups' images with 32-bit offsets)
                ws = _ws(nbytes, x.device, slot=1)
                _launch("conv64_wino_wgrad_kernel", _conv64_key(d, "wgrad"), _conv64_flop(d),
                        lambda: C.conv64_wino_bwd_weight(ptr(x), ptr(dy), ptr(dw), ptr(ws), nbytes, d, stream()))
            else:
                nbytes = C.conv64_bwd_weight_workspace(d)
                ws = _ws(nbytes, x.device, slot=1)
                _launch("conv64_wgrad_kernel", _conv64_key(d, "wgrad"), _conv64_flop(d),
                        lambda: C.conv64_bwd_weight(ptr(x), ptr(dy), ptr(dw), ptr(db), None, None, ptr(ws), nbytes, d,
                                                    stream()))
        dx = None
        if ctx.needs_dx:
            dx = torch.empty_like(x)
            link = ctx.in_link
            rec = link.record if (link is not None and _POOL_LINK) else None
            if rec is not None and rec[0].data_ptr() == x.data_ptr():
                # dx = d(pooled): the pooled block's BatchNorm-backward sums come out of this launch's epilogue
                pooled, pbnp, py, pargmax, pd = rec
                if ctx.wino:
                    part = torch.empty((C.conv64_wino_bwd_data_rows(d), 128), dtype=torch.float32, device=x.device)
                    _launch("conv64_wino_kernel", _conv64_key(d, "dgrad"), _conv64_flop(d),
                            lambda: C.conv64_wino_bwd_data_pool_sums(ptr(dy), ptr(packs[1]), ptr(dx), ptr(pooled), ptr(pbnp), ptr(py),
                                                                     ptr(pargmax), pd, ptr(part), d, stream()))
                else:
                    part = torch.empty((C.conv64_bwd_data_tiles(d), 128), dtype=torch.float32, device=x.device)
                    _launch("conv64_dgrad_poolsum_kernel", _conv64_key(d, "dgrad"), _conv64_flop(d),
                            lambda: C.conv64_bwd_data_pool_sums(ptr(dy), ptr(packs[1]), ptr(dx), ptr(pooled), ptr(pbnp), ptr(py),
                                                                ptr(pargmax), pd, ptr(part), d, stream()))
                link.partials = part
            elif ctx.wino:
                _launch("conv64_wino_kernel", _conv64_key(d, "dgrad"), _conv64_flop(d),
                        lambda: C.conv64_wino_bwd_data(ptr(dy), ptr(packs[1]), ptr(dx), d, stream()))
            else:
                _launch("conv64_gather_pipe_kernel" if C.conv64_gather_pipe_supported(d, 1) else "conv64_fwd_kernel",
                        _conv64_key(d, "dgrad"), _conv64_flop(d),
                        lambda: C.conv64_bwd_data(ptr(dy), ptr(packs[1]), ptr(dx), None, d, stream()))
        return dx, _give(ctx.params[0], dw), _give(ctx.params[1], db), None, None, None, None, None


# ----------------------------------------------------------------------------------------------------------------
# BatchNorm2d + ReLU (+ MaxPool2d(3, 2, pad)) — models/models.py:50-52,55-57,60-62 / 67-68,71-72,75-76,79-80
# ----------------------------------------------------------------------------------------------------------------
# When the two frames of a step are encoded on two different streams (learner.trainStep), the momentum updates of one
# BatchNorm layer's running statistics must still happen in program order (obs, then next_obs): every update waits for
# the event the previous update of the same buffers recorded.
_bn_last_update = {}


def _ordered_bn_update(running_mean, fn):
    """Run `fn` (a launch that updates the running statistics of one layer — or of several: a list) behind the previous update of the
    same buffers, whichever stream that was enqueued on."""
    rms = running_mean if isinstance(running_mean, (list, tuple)) else [running_mean]
    cur = torch.cuda.current_stream(rms[0].device)
    for rm in rms:
        prev = _bn_last_update.get(rm.data_ptr())
        if prev is not None and prev[0] != cur.cuda_stream:
            cur.wait_event(prev[1])
    fn()
    ev = torch.cuda.Event()
    ev.record(cur)
    for rm in rms:
        _bn_last_update[rm.data_ptr()] = (cur.cuda_stream, ev)


def _bn_params(stats, count, gamma, beta, running_mean, running_var, training, device, tick=None):
    """BatchNorm record(s) of a layer: training -> one 256-float record per group from the producing convolution's
    per-tile partials (`count` = positions of ALL groups together), eval -> one record from the running statistics."""
    groups = cur_groups(training)
    if tick is None:
        tick = getattr(running_mean, "_srlz_tick", None)  # num_batches_tracked of the layer (hotpath._bn_args)
    bnp = torch.empty(256 * groups, dtype=torch.float32, device=device)
    batch_stat = None
    if training:
        batch_stat = torch.empty(128 * groups, dtype=torch.float32, device=device)
        nbytes = C.bn_bwd_workspace(0)
        ws = _ws(nbytes, device)

        def update():  # (tick = num_batches_tracked: advanced by `groups` inside the same launch)
            C.bn_finalize(ptr(stats), stats.shape[0], groups, count // groups, ptr(gamma), ptr(beta), BN_EPS, BN_MOMENTUM, 1,
                          ptr(running_mean), ptr(running_var), ptr(tick), ptr(bnp), ptr(batch_stat), ptr(ws), nbytes, stream())
        _ordered_bn_update(running_mean, update)
    else:
        C.bn_eval_params(ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), BN_EPS, ptr(bnp), stream())
    return bnp, batch_stat


class BNReLUPoolFn(Function):
    """y (raw conv output, NHWC) -> maxpool(relu(bn(y))).  `stats` are the conv's per-tile partial sums."""

    @staticmethod
    def forward(ctx, y, stats, gamma, beta, running_mean, running_var, training, pool_pad, out_nchw, stat_sink, out_link=None):
        y = _check(y, "bn input")
        n, h, w, _ = y.shape
        hp, wp = (h + 2 * pool_pad - 3) // 2 + 1, (w + 2 * pool_pad - 3) // 2 + 1
        d = PoolDesc(n, h, w, hp, wp, pool_pad, 1 if out_nchw else 0, cur_groups(training))
        bnp, batch_stat = _bn_params(stats, n * h * w, gamma, beta, running_mean, running_var, training, y.device)
        if stat_sink is not None and batch_stat is not None:
            stat_sink.append(batch_stat)
        shape = (n, 64, hp, wp) if out_nchw else (n, hp, wp, 64)
        pooled = torch.empty(shape, dtype=torch.float32, device=y.device)
        need_bwd = ctx.needs_input_grad[0] or ctx.needs_input_grad[2]
        argmax = torch.empty((n, hp, wp, 64), dtype=torch.uint8, device=y.device) if need_bwd else None
        C.bn_relu_pool_fwd(ptr(y), ptr(bnp), ptr(pooled), ptr(argmax), d, stream())
        if need_bwd:
            ctx.save_for_backward(y, bnp, argmax, pooled)
            if out_link is not None and not out_nchw:
                out_link.record = (pooled, bnp, y, argmax, d)
        ctx.desc, ctx.training, ctx.out_link = d, training, out_link
        ctx.params = (gamma, beta)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        y, bnp, argmax, pooled = ctx.saved_tensors
        dpooled = _check(dpooled, "pool grad")
        dy = torch.empty_like(y)
        dgamma = _gbuf(ctx.params[0])
        dbeta = _gbuf(ctx.params[1])
        part = ctx.out_link.take_partials() if ctx.out_link is not None else None
        nbytes = C.bn_bwd_workspace(0)
        ws = _ws(nbytes, y.device)
        if part is not None:  # the consumer's data gradient took the two sums in its epilogue (PoolLink)
            sums = torch.empty(128 * max(ctx.desc.groups, 1), dtype=torch.float32, device=y.device)
            C.bn_bwd_finalize_partials(ptr(part), part.shape[0], max(ctx.desc.groups, 1), ptr(sums), ptr(dgamma), ptr(dbeta), ptr(ws),
                                       nbytes, stream())
            C.bn_relu_pool_bwd_apply(ptr(y), ptr(bnp), ptr(argmax), ptr(dpooled), ptr(sums), ptr(dy), 1 if ctx.training else 0,
                                     ctx.desc, stream())
        else:
            C.bn_relu_pool_bwd(ptr(y), ptr(bnp), ptr(argmax), ptr(dpooled), ptr(pooled), ptr(dy), ptr(dgamma), ptr(dbeta),
                               1 if ctx.training else 0, ptr(ws), nbytes, ctx.desc, stream())
        return dy, None, _give(ctx.params[0], dgamma), _give(ctx.params[1], dbeta), None, None, None, None, None, None, None


class EncInFn(Function):
    """First encoder block as ONE autograd node: Conv2d(C,64,7,2,3,bias=False) -> BatchNorm2d -> ReLU -> MaxPool(3,2,1)
    (models/models.py:49-52).  Forward = Conv1Fn + BNReLUPoolFn; backward never writes d(loss)/d(conv output): the
    weight-gradient kernel rebuilds it from (y, argmax, dpooled, BN sums) while staging its operand
    (srlz_conv1_bwd_weight_fused).  Saved tensors start with (y, bnp, argmax) like BNReLUPoolFn's."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, running_mean, running_var, training, pool_pad, stat_sink, out_link=None):
        u8 = is_u8_frames(x)  # the loader's bytes: normalised inside the kernels' window staging
        x, w = (_check_u8(x, "conv1 input") if u8 else _check(x, "conv1 input")), _check(w, "conv1 weight")
        n, c, h, wd = x.shape
        d = _skinny_desc(n, c, h, wd, 0, cur_groups(training))
        y = torch.empty((n, d.hf, d.wf, 64), dtype=torch.float32, device=x.device)
        stats = torch.empty((C.skinny_tiles(d), 128), dtype=torch.float32, device=x.device) if training else None
        if u8:
            lut = norm_lut(x.device)
            _launch("skinny_conv_kernel", _skinny_key(d, "fwd u8"), _conv1_flop(d),
                    lambda: C.conv1_fwd_u8(ptr(x), ptr(lut), ptr(w), ptr(y), ptr(stats), d, stream()))
        else:
            _launch("skinny_conv_kernel", _skinny_key(d, "fwd"), _conv1_flop(d),
                    lambda: C.conv1_fwd(ptr(x), ptr(w), ptr(y), ptr(stats), d, stream()))
        hp, wp = (d.hf + 2 * pool_pad - 3) // 2 + 1, (d.wf + 2 * pool_pad - 3) // 2 + 1
        pd = PoolDesc(n, d.hf, d.wf, hp, wp, pool_pad, 0, cur_groups(training))
        bnp, batch_stat = _bn_params(stats, n * d.hf * d.wf, gamma, beta, running_mean, running_var, training, x.device)
        if stat_sink is not None and batch_stat is not None:
            stat_sink.append(batch_stat)
        pooled = torch.empty((n, hp, wp, 64), dtype=torch.float32, device=x.device)
        need_bwd = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        argmax = torch.empty((n, hp, wp, 64), dtype=torch.uint8, device=x.device) if need_bwd else None
        C.bn_relu_pool_fwd(ptr(y), ptr(bnp), ptr(pooled), ptr(argmax), pd, stream())
        if need_bwd:
            ctx.save_for_backward(y, bnp, argmax, pooled, x, w)
            if out_link is not None:
                out_link.record = (pooled, bnp, y, argmax, pd)
        ctx.out_link = out_link
        ctx.desc, ctx.pdesc, ctx.training = d, pd, training
        ctx.params = (gamma, beta)
        ctx.mark_non_differentiable(y)
        ctx.set_materialize_grads(False)  # no 0.8 GB zero tensor for the (non-differentiable) second output
        return pooled, y

    @staticmethod
    def backward(ctx, dpooled, _dy):
        y, bnp, argmax, pooled, x, w = ctx.saved_tensors
        dpooled = _check(dpooled, "pool grad")
        dev = y.device
        dgamma = _gbuf(ctx.params[0])
        dbeta = _gbuf(ctx.params[1])
        sums = torch.empty(128 * max(ctx.pdesc.groups, 1), dtype=torch.float32, device=dev)
        nbytes = C.bn_bwd_workspace(0)
        ws = _ws(nbytes, dev)
        part = ctx.out_link.take_partials() if ctx.out_link is not None else None
        if part is not None:  # conv2's data gradient took the two sums in its epilogue (PoolLink)
            C.bn_bwd_finalize_partials(ptr(part), part.shape[0], max(ctx.pdesc.groups, 1), ptr(sums), ptr(dgamma), ptr(dbeta), ptr(ws),
                                       nbytes, stream())
        else:
            C.bn_relu_pool_bwd_sums(ptr(y), ptr(bnp), ptr(argmax), ptr(dpooled), ptr(pooled), ptr(sums), ptr(dgamma), ptr(dbeta),
                                    ptr(ws), nbytes, ctx.pdesc, stream())
        dw = _gbuf(w)
        nbytes = C.skinny_bwd_weight_workspace(ctx.desc)
        ws = _ws(nbytes, dev)
        if is_u8_frames(x):
            lut = norm_lut(dev)
            _launch("skinny_wgrad_kernel", _skinny_key(ctx.desc, "wgrad fused u8"), _conv1_flop(ctx.desc),
                    lambda: C.conv1_bwd_weight_fused_u8(ptr(x), ptr(lut), ptr(y), ptr(bnp), ptr(argmax), ptr(dpooled), ptr(sums),
                                                        1 if ctx.training else 0, ptr(dw), ptr(ws), nbytes, ctx.desc, ctx.pdesc,
                                                        stream()))
        else:
            _launch("skinny_wgrad_kernel", _skinny_key(ctx.desc, "wgrad fused"), _conv1_flop(ctx.desc),
                    lambda: C.conv1_bwd_weight_fused(ptr(x), ptr(y), ptr(bnp), ptr(argmax), ptr(dpooled), ptr(sums),
                                                     1 if ctx.training else 0, ptr(dw), ptr(ws), nbytes, ctx.desc, ctx.pdesc,
                                                     stream()))
        return None, _give(w, dw), _give(ctx.params[0], dgamma), _give(ctx.params[1], dbeta), None, None, None, None, None, None


# ----------------------------------------------------------------------------------------------------------------
# Decoder blocks with the BatchNorm-apply + ReLU fused into the NEXT layer's operand load
# (models/models.py:67-82: BatchNorm2d -> ReLU -> ConvTranspose2d).  Input = RAW output of the previous transposed
# convolution + its BatchNorm statistics; the activated tensor relu(bn(y)) is never written to memory.
# ----------------------------------------------------------------------------------------------------------------
def _groups_of(bnp):
    """Number of BatchNorm groups a record tensor holds (256 floats per group)."""
    return bnp.numel() // 256


def _bn_relu_backward(y, bnp, da, training, gb=(None, None)):
    dy = torch.empty_like(y)
    dgamma = _gbuf(gb[0], 64, y.device)
    dbeta = _gbuf(gb[1], 64, y.device)
    nbytes = C.bn_bwd_workspace(0)
    ws = _ws(nbytes, y.device)
    C.bn_relu_bwd(ptr(y), ptr(bnp), ptr(da), ptr(dy), ptr(dgamma), ptr(dbeta), 1 if training else 0, ptr(ws), nbytes,
                  y.numel() // 64, _groups_of(bnp), stream())
    return dy, dgamma, dbeta


class BwdLink:
    """Hand-over of a DEFERRED BatchNorm+ReLU backward between two consecutive decoder blocks.

    Block k+1 (consumer of y_k) owns the backward of BatchNorm_k + ReLU.  Instead of materialising d(loss)/dy_k it only
    computes the two BatchNorm sums, leaves (y_k, bnp_k, sums_k) here and hands dA_k back through autograd; block k
    (producer of y_k, the ONLY consumer of that gradient) picks the record up in its own backward and lets its two
    kernels rebuild d(loss)/dy_k while loading their operand (srlz_bn_bwd_operand).  A link is created per forward
    pass by hotpath.decoder_forward, connects exactly one producer with one consumer, and is only used when nothing
    else looks at y_k's gradient (hotpath.TAPS off)."""

    def __init__(self):
        self.record = None

    def put(self, y, bnp, sums, training):
        self.record = (y, bnp, sums, training)

    def take(self):
        rec, self.record = self.record, None
        return rec


def _bn_relu_backward_sums(y, bnp, da, gb=(None, None)):
    dev = y.device
    groups = _groups_of(bnp)
    sums = torch.empty(128 * groups, dtype=torch.float32, device=dev)
    dgamma = _gbuf(gb[0], 64, dev)
    dbeta = _gbuf(gb[1], 64, dev)
    nbytes = C.bn_bwd_workspace(0)
    ws = _ws(nbytes, dev)
    C.bn_relu_bwd_sums(ptr(y), ptr(bnp), ptr(da), ptr(sums), ptr(dgamma), ptr(dbeta), ptr(ws), nbytes, y.numel() // 64,
                       groups, stream())
    return sums, dgamma, dbeta


def _bn_backward_for_producer(link, y_prev, bnp, da, training, partial=None, gb=(None, None)):
    """BatchNorm+ReLU backward of the block input: deferred to the producer through `link`, or materialised.
    `partial`: per-tile partial sums already written by the epilogue of the kernel that produced `da`."""
    if link is not None:
        if partial is not None:
            sums, dgamma, dbeta = _bn_backward_sums_from_partials(partial, gb, _groups_of(bnp))
        else:
            sums, dgamma, dbeta = _bn_relu_backward_sums(y_prev, bnp, da, gb)
        link.put(y_prev, bnp, sums, training)
        return da, dgamma, dbeta
    return _bn_relu_backward(y_prev, bnp, da, training, gb)


def _bn_backward_sums_from_partials(partial, gb=(None, None), groups=1):
    dev = partial.device
    sums = torch.empty(128 * groups, dtype=torch.float32, device=dev)
    dgamma = _gbuf(gb[0], 64, dev)
    dbeta = _gbuf(gb[1], 64, dev)
    nbytes = C.bn_bwd_workspace(0)
    ws = _ws(nbytes, dev)
    C.bn_bwd_finalize_partials(ptr(partial), partial.shape[0], groups, ptr(sums), ptr(dgamma), ptr(dbeta), ptr(ws), nbytes,
                               stream())
    return sums, dgamma, dbeta


def _operand_from(link, materialise=True):
    """(ctypes record or None, dy_out or None, keep-alive tuple) for the gradient arriving at a producer block.
    materialise: the record asks the data-gradient kernel to also store the rebuilt d(loss)/dy (dy_out) for a separate
    weight-gradient launch; False when one kernel consumes the operand for both gradients (srlz_conv64_bwd_fused)."""
    rec = link.take() if link is not None else None
    if rec is None:
        return None, None, None
    y, bnp, sums, training = rec
    dy_out = torch.empty_like(y) if materialise else None
    op = C.BnBwdOperand(y.data_ptr(), bnp.data_ptr(), sums.data_ptr(), y.numel() // 64 // _groups_of(bnp), 1 if training else 0,
                        dy_out.data_ptr() if materialise else None)
    return op, dy_out, rec


# Data + weight + bias gradient of a decoder block's ConvTranspose are ONE launch wherever the shape allows
# (srlz_conv64_bwd_fused_supported: at least 8 tiles, a low-resolution grid of at most 63 columns, two BatchNorm groups at most);
# other shapes take a data-gradient launch that stores the rebuilt d(loss)/dy and a weight-gradient launch that reads it back.


class DecBlockFn(Function):
    """(y_prev raw, stats_prev, gamma, beta, running stats) -> y = ConvTranspose2d(64,64,3,2)(relu(bn(y_prev))) + stats.
    in_link: BwdLink shared with the producer of y_prev (or None); out_link: shared with the consumer of y (or None)."""

    @staticmethod
    def forward(ctx, y_prev, stats_prev, gamma, beta, running_mean, running_var, training, w, bias, want_stats,
                in_link=None, out_link=None):
        y_prev, w = _check(y_prev, "decoder block input"), _check(w, "convT weight")
        n, hi, wi, _ = y_prev.shape
        bnp, _ = _bn_params(stats_prev, n * hi * wi, gamma, beta, running_mean, running_var, training, y_prev.device)
        d = conv64_desc(n, hi, wi, 2, 0, True, cur_groups(training))
        packs = torch.empty((2, C.conv64_packed_floats()), dtype=torch.float32, device=y_prev.device)
        C.conv64_pack_weights(ptr(w), ptr(packs[0]), ptr(packs[1]), d, stream())
        y = torch.empty((n, d.ho, d.wo, 64), dtype=torch.float32, device=y_prev.device)
        stats = torch.empty((C.conv64_fwd_tiles(d), 128), dtype=torch.float32, device=y_prev.device) if want_stats else None
        _launch("conv64_fwd_kernel", _conv64_key(d, "fwd"), _conv64_flop(d),
                lambda: C.conv64_fwd(ptr(y_prev), ptr(packs[0]), ptr(bias), ptr(y), ptr(stats), ptr(bnp), d, stream()))
        ctx.save_for_backward(y_prev, bnp, packs)
        ctx.desc, ctx.training = d, training
        ctx.in_link, ctx.out_link = in_link, out_link
        ctx.params = (gamma, beta, w, bias)
        if stats is None:
            stats = torch.empty(0, device=y_prev.device)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)  # no zero tensor for the statistics output in backward
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        y_prev, bnp, packs = ctx.saved_tensors
        d = ctx.desc
        dy = _check(dy, "decoder block dy")
        if ctx.out_link is not None and ctx.out_link.record is not None and C.conv64_bwd_fused_supported(d):
            # `dy` is dA of the following BatchNorm+ReLU: ONE kernel rebuilds the true dy while staging it and contracts it both ways
            # (data gradient and weight / bias gradient) — it is never written to memory
            dy_bn, _, keep = _operand_from(ctx.out_link, materialise=False)
            da = torch.empty_like(y_prev)
            dw = _gbuf(ctx.params[2])
            db = _gbuf(ctx.params[3], 64, dy.device)
            nbytes = C.conv64_bwd_fused_workspace(d)
            ws = _ws(nbytes, dy.device, slot=1)
            # with the producer's BatchNorm backward deferred (in_link), its two sums come out of this launch's flush as partial records
            part = torch.empty((C.conv64_bwd_fused_bn_rows(d), 128), dtype=torch.float32, device=dy.device) if ctx.in_link is not None else None
            _launch("conv64_bwd_fused_kernel", _conv64_key(d, "dgrad+wgrad"), 2.0 * _conv64_flop(d),
                    lambda: C.conv64_bwd_fused(ptr(y_prev), ptr(bnp), ptr(dy), dy_bn, ptr(packs[1]), ptr(da), ptr(dw), ptr(db), ptr(part),
                                               ptr(ws), nbytes, d, stream()))
            dy_prev, dgamma, dbeta = _bn_backward_for_producer(ctx.in_link, y_prev, bnp, da, ctx.training, partial=part, gb=ctx.params[:2])
            gp, bp, wp, cp = ctx.params
            return dy_prev, None, _give(gp, dgamma), _give(bp, dbeta), None, None, None, _give(wp, dw), _give(cp, db), None, None, None
        # dy_bn not None: `dy` is dA of the following BatchNorm+ReLU; the data-gradient kernel rebuilds the true dy in its
        # operand load and stores it (dy_true) for the weight-gradient kernel, which therefore runs second
        dy_bn, dy_true, keep = _operand_from(ctx.out_link)
        da = torch.empty_like(y_prev)
        # (the fused-operand launch is a different instantiation of the kernel: timed under its own name)
        _launch("conv64_fwd_kernel" if dy_bn is None else "conv64_fwd_kernel<bn-bwd operand>", _conv64_key(d, "dgrad"),
                _conv64_flop(d), lambda: C.conv64_bwd_data(ptr(dy), ptr(packs[1]), ptr(da), dy_bn, d, stream()))
        if dy_true is not None:
            dy = dy_true
        dw = _gbuf(ctx.params[2])
        db = _gbuf(ctx.params[3], 64, dy.device)
        nbytes = C.conv64_bwd_weight_workspace(d)
        ws = _ws(nbytes, dy.device, slot=1)
        _launch("conv64_wgrad_kernel", _conv64_key(d, "wgrad"), _conv64_flop(d),
                lambda: C.conv64_bwd_weight(ptr(y_prev), ptr(dy), ptr(dw), ptr(db), ptr(bnp), None, ptr(ws), nbytes, d,
                                            stream()))
        dy_prev, dgamma, dbeta = _bn_backward_for_producer(ctx.in_link, y_prev, bnp, da, ctx.training, gb=ctx.params[:2])
        gp, bp, wp, cp = ctx.params
        return dy_prev, None, _give(gp, dgamma), _give(bp, dbeta), None, None, None, _give(wp, dw), _give(cp, db), None, None, None


# The last ConvTranspose's backward is ONE fused launch whenever the BatchNorm backward in front of it is deferred (in_link) and the
# shape is supported (C = 3 / 6); the two-launch form serves hotpath.TAPS (no link) and is the reference of the fused kernel's tests.


class DecOutFn(Function):
    """(y_prev raw, stats, BN params) -> ConvTranspose2d(64, C, 4, 2)(relu(bn(y_prev))), NCHW (models/models.py:79-82)."""

    @staticmethod
    def forward(ctx, y_prev, stats_prev, gamma, beta, running_mean, running_var, training, w, bias, in_link=None):
        y_prev, w = _check(y_prev, "decoder output input"), _check(w, "convT_out weight")
        n, hf, wf, _ = y_prev.shape
        bnp, _ = _bn_params(stats_prev, n * hf * wf, gamma, beta, running_mean, running_var, training, y_prev.device)
        c = w.shape[1]
        d = SkinnyDesc(n, c, (hf - 1) * 2 + 4, (wf - 1) * 2 + 4, hf, wf, 1, cur_groups(training))
        y = torch.empty((n, c, d.himg, d.wimg), dtype=torch.float32, device=y_prev.device)
        _launch("convT_out_os_kernel", _skinny_key(d, "fwd"), _convT_out_flop(d),
                lambda: C.convT_out_fwd(ptr(y_prev), ptr(w), ptr(bias), ptr(y), ptr(bnp), d, stream()))
        ctx.save_for_backward(y_prev, bnp, w)
        ctx.desc, ctx.training, ctx.in_link = d, training, in_link
        ctx.params = (gamma, beta, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        y_prev, bnp, w = ctx.saved_tensors
        dy = _check(dy, "decoder output dy")
        return _dec_out_backward(ctx, y_prev, bnp, w, dy, None) + (None,)


def _dec_out_backward(ctx, y_prev, bnp, w, dy, gain):
    """Backward of the last ConvTranspose (DecOutFn / DecOutLossFn): gradients for (y_prev, stats, gamma, beta, rm, rv, training,
    w, bias).  gain = (upstream scalar tensor, div, coef): `dy` holds the reconstruction error and the loss gradient is
    ((upstream / div) * coef) * dy — applied inside the fused kernel, or materialised in place for the two-launch path."""
    d = ctx.desc
    dw = _gbuf(w)
    db = _gbuf(ctx.params[2], d.c, dy.device)
    if ctx.in_link is not None and C.convT_out_bwd_fused_supported(d):
        # data gradient, its BatchNorm-backward partials and the weight / bias gradients in one pass over (dy, y_prev)
        da = torch.empty_like(y_prev)
        partial = torch.empty((C.convT_out_bwd_fused_tiles(d), 128), dtype=torch.float32, device=dy.device)
        nbytes = C.convT_out_bwd_fused_workspace(d)
        ws = _ws(nbytes, dy.device, slot=1)
        g_dev, g_div, g_coef = (ptr(gain[0]), gain[1], gain[2]) if gain is not None else (None, 1.0, 1.0)
        _launch("convT_out_os_bwd_kernel", _skinny_key(d, "dgrad+wgrad"), 2.0 * _convT_out_flop(d),
                lambda: C.convT_out_bwd_fused(ptr(dy), ptr(w), ptr(da), ptr(y_prev), ptr(bnp), ptr(partial), ptr(dw), ptr(db), ptr(ws),
                                              nbytes, g_dev, g_div, g_coef, d, stream()))
        dy_prev, dgamma, dbeta = _bn_backward_for_producer(ctx.in_link, y_prev, bnp, da, ctx.training, partial, gb=ctx.params[:2])
        gp, bp, cp = ctx.params
        return dy_prev, None, _give(gp, dgamma), _give(bp, dbeta), None, None, None, _give(w, dw), _give(cp, db)
    if gain is not None:  # the two-launch kernels take the gradient itself: error -> gradient, in place (the error has no other use)
        C.scale_by_scalar(ptr(dy), ptr(gain[0]), gain[1], gain[2], ptr(dy), dy.numel(), stream())
    nbytes = C.skinny_bwd_weight_workspace(d)
    ws = _ws(nbytes, dy.device, slot=1)
    C.convT_out_bwd_weight(ptr(y_prev), ptr(dy), ptr(dw), ptr(db), ptr(bnp), ptr(ws), nbytes, d, stream())
    da = torch.empty_like(y_prev)
    # with the BatchNorm backward deferred (in_link), its two sums come out of this kernel's epilogue
    partial = None
    if ctx.in_link is not None:
        partial = torch.empty((C.skinny_tiles(d), 128), dtype=torch.float32, device=dy.device)
    C.convT_out_bwd_data(ptr(dy), ptr(w), ptr(da), ptr(y_prev) if partial is not None else None,
                         ptr(bnp) if partial is not None else None, ptr(partial), d, stream())
    dy_prev, dgamma, dbeta = _bn_backward_for_producer(ctx.in_link, y_prev, bnp, da, ctx.training, partial, gb=ctx.params[:2])
    gp, bp, cp = ctx.params
    return dy_prev, None, _give(gp, dgamma), _give(bp, dbeta), None, None, None, _give(w, dw), _give(cp, db)


class DecOutLossFn(Function):
    """The last ConvTranspose AND the step's reconstruction / generation loss as one node (K11 / K12 of SURVEY.md 8a'; reference
    models/models.py:79-82 followed by losses/losses.py:172-214): (y_prev raw, stats, BN params, weights, target [obs ; next_obs])
    -> (loss scalar, err = dec - target [not differentiable; the backward's operand]).  mean: reconstructionLoss x 2
    (sum / numel per frame, added), else F.mse_loss(reduction='sum') x 2.  The batch is the pair of frames of a step.
    Backward: d(loss)/d(dec) = ((upstream / div) * 2) * err, formed inside the ConvTranspose's backward kernel."""

    @staticmethod
    def forward(ctx, y_prev, stats_prev, gamma, beta, running_mean, running_var, training, w, bias, in_link, target, mean):
        y_prev, w = _check(y_prev, "decoder output input"), _check(w, "convT_out weight")
        u8 = is_u8_frames(target)
        target = _check_u8(target, "reconstruction target") if u8 else _check(target, "reconstruction target")
        n, hf, wf, _ = y_prev.shape
        bnp, _ = _bn_params(stats_prev, n * hf * wf, gamma, beta, running_mean, running_var, training, y_prev.device)
        c = w.shape[1]
        d = SkinnyDesc(n, c, (hf - 1) * 2 + 4, (wf - 1) * 2 + 4, hf, wf, 1, cur_groups(training))
        if tuple(target.shape) != (n, c, d.himg, d.wimg) or n % 2:
            raise C.SrlzError("fused reconstruction loss: target %s does not match the decoder output %s"
                              % (tuple(target.shape), (n, c, d.himg, d.wimg)))
        err = torch.empty((n, c, d.himg, d.wimg), dtype=torch.float32, device=y_prev.device)
        nwg = C.convT_out_fwd_loss_workgroups(d)
        part = _ws(2 * nwg * 8, y_prev.device, slot=2)
        if u8:
            lut = norm_lut(y_prev.device)
            _launch("convT_out_os_kernel", _skinny_key(d, "fwd+loss u8"), _convT_out_flop(d),
                    lambda: C.convT_out_fwd_loss_u8(ptr(y_prev), ptr(w), ptr(bias), ptr(target), ptr(lut), ptr(err), None, ptr(bnp),
                                                    ptr(part), d, stream()))
        else:
            _launch("convT_out_os_kernel", _skinny_key(d, "fwd+loss"), _convT_out_flop(d),
                    lambda: C.convT_out_fwd_loss(ptr(y_prev), ptr(w), ptr(bias), ptr(target), ptr(err), None, ptr(bnp), ptr(part), d,
                                                 stream()))
        sums = torch.empty(2, dtype=torch.float32, device=y_prev.device)
        comb = torch.empty((), dtype=torch.float32, device=y_prev.device)
        per_frame = err.numel() // 2
        C.pair_loss_finalize(ptr(part), nwg, per_frame, 1 if mean else 0, ptr(sums), ptr(comb), stream())
        ctx.save_for_backward(y_prev, bnp, w, err)
        ctx.desc, ctx.training, ctx.in_link = d, training, in_link
        ctx.params = (gamma, beta, bias)
        ctx.div = float(per_frame) if mean else 1.0
        ctx.mark_non_differentiable(err)
        ctx.set_materialize_grads(False)
        return comb, err

    @staticmethod
    def backward(ctx, g, _derr):
        y_prev, bnp, w, err = ctx.saved_tensors
        g = _check(g, "loss grad")
        return _dec_out_backward(ctx, y_prev, bnp, w, err, (g, ctx.div, 2.0)) + (None, None, None)


def bn_relu_materialise(y, bnp_source):
    """relu(bn(y)) as a tensor — debug / test helper only (the product path fuses it into the consumer)."""
    a = torch.empty_like(y)
    C.bn_relu_fwd(ptr(y), ptr(bnp_source), ptr(a), y.numel() // 64, stream())
    return a


# ----------------------------------------------------------------------------------------------------------------
# nn.Linear — autoencoders.py:94-100, vae.py:52-57, forward_inverse.py:16,48-55
# ----------------------------------------------------------------------------------------------------------------
class LinearFn(Function):
    @staticmethod
    def forward(ctx, x, w, b, relu):
        x, w = _check(x, "linear input"), _check(w, "linear weight")
        m, k = x.shape
        n = w.shape[0]
        assert w.shape[1] == k
        y = torch.empty((m, n), dtype=torch.float32, device=x.device)
        nbytes = C.linear_workspace(m, n, k)
        ws = _ws(nbytes, x.device)
        C.linear_fwd(ptr(x), ptr(w), ptr(b), ptr(y), m, n, k, 1 if relu else 0, ptr(ws), nbytes, stream())
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.relu, ctx.has_bias, ctx.needs_dx = relu, b is not None, ctx.needs_input_grad[0]
        ctx.params = (b,)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = _check(dy, "linear dy")
        m, k = x.shape
        n = w.shape[0]
        if ctx.relu:
            dy = dy.clone()
            C.relu_bwd_inplace(ptr(y), ptr(dy), dy.numel(), stream())
        nbytes = C.linear_workspace(m, n, k)
        ws = _ws(nbytes, x.device)
        dx = None
        if ctx.needs_dx:
            dx = torch.empty_like(x)
            C.linear_bwd_data(ptr(dy), ptr(w), ptr(dx), m, n, k, ptr(ws), nbytes, stream())
        dw = _gbuf(w)
        db = _gbuf(ctx.params[0]) if ctx.has_bias else None
        C.linear_bwd_weight(ptr(dy), ptr(x), ptr(dw), ptr(db), m, n, k, ptr(ws), nbytes, stream())
        return dx, _give(w, dw), _give(ctx.params[0], db), None


class TotalLossFn(Function):
    """total = sum_i w_i * l_i over 0-dim device scalars, as LossManager.computeTotalLoss forms it (reference losses/losses.py:55-56:
    Python's left-to-right sum of separately rounded fp32 products), in ONE launch that also drops [total, l_0, l_1, ...] into
    `tail` (the gradient bucket's scalar tail) when given; backward: d l_i = w_i * d total, one launch."""

    @staticmethod
    def forward(ctx, weights, tail, *losses):
        import ctypes
        n = len(losses)
        losses = [_check(l, "loss term").reshape(()) for l in losses]
        ptrs = (ctypes.c_void_p * n)(*[l.data_ptr() for l in losses])
        w = (ctypes.c_float * n)(*[float(x) for x in weights])
        total = torch.empty((), dtype=torch.float32, device=losses[0].device)
        C.weighted_total(ptrs, w, n, ptr(total), ptr(tail), stream())
        ctx.weights, ctx.n = w, n
        ctx.keep = losses  # (the kernel is asynchronous: the terms must outlive it)
        return total

    @staticmethod
    def backward(ctx, dout):
        g = torch.empty(ctx.n, dtype=torch.float32, device=dout.device)
        C.weighted_total_bwd(ptr(dout.contiguous()), ctx.weights, ctx.n, ptr(g), stream())
        return (None, None) + tuple(g[i] for i in range(ctx.n))


class MaskColumnsFn(Function):
    """x with every column outside [lo, hi) zeroed — SRLModulesSplit.detachSplit (reference models/modules.py:191-236)
    rebuilds the state from th.zeros_like blocks and one kept slice, i.e. applies this mask; gradient = same mask."""

    @staticmethod
    def forward(ctx, x, lo, hi):
        x = _check(x, "split input")
        rows, cols = x.shape
        y = torch.empty_like(x)
        C.mask_columns(ptr(x), ptr(y), rows, cols, lo, hi, stream())
        ctx.range = (lo, hi)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _check(dy, "split grad")
        rows, cols = dy.shape
        dx = torch.empty_like(dy)
        C.mask_columns(ptr(dy), ptr(dx), rows, cols, ctx.range[0], ctx.range[1], stream())
        return dx, None, None


_param_tables = {}


def _param_table(params):
    """Device arrays (addresses, lengths) of a parameter list, cached per list of storage addresses."""
    key = tuple((p.data_ptr(), p.numel()) for p in params)
    tab = _param_tables.get(key)
    if tab is None:
        dev = params[0].device
        tab = (torch.tensor([k[0] for k in key], dtype=torch.int64, device=dev),
               torch.tensor([k[1] for k in key], dtype=torch.int64, device=dev))
        _param_tables.clear()  # parameters are re-homed rarely (FlatParams): keep one table
        _param_tables[key] = tab
    return tab


class ParamNormFn(Function):
    """mode 0: sum_i sum|p_i| (l1Loss, reference losses.py:132-142); mode 1: (sum_i ||p_i||_2) / len(params) (l2Loss,
    losses.py:145-155).  One launch over the whole parameter list (one workgroup per tensor, fp64 accumulation)."""

    @staticmethod
    def forward(ctx, mode, *params):
        for p in params:
            _check(p, "regularised parameter")
        nseg = len(params)
        ptrs, lens = _param_table(params)
        dev = params[0].device
        norms = torch.empty(nseg, dtype=torch.float32, device=dev)
        out = torch.empty((), dtype=torch.float32, device=dev)
        scale = 1.0 if mode == 0 else 1.0 / nseg
        C.param_norms(ptr(ptrs), ptr(lens), nseg, mode, scale, ptr(norms), ptr(out), stream())
        ctx.save_for_backward(norms, ptrs, lens, *params)
        ctx.mode, ctx.scale = mode, scale
        return out

    @staticmethod
    def backward(ctx, dout):
        norms, ptrs, lens = ctx.saved_tensors[:3]
        params = ctx.saved_tensors[3:]
        dout = dout.contiguous()
        grads = [_gbuf(p) for p in params]  # staging views for bucketed parameters: fixed addresses, table cached
        key = tuple(g.data_ptr() for g in grads)
        gptrs = _grad_tables.get(key)
        if gptrs is None:
            gptrs = torch.tensor(key, dtype=torch.int64, device=norms.device)
            _grad_tables.clear()
            _grad_tables[key] = gptrs
        C.param_norms_grad(ptr(ptrs), ptr(gptrs), ptr(lens), len(params), ctx.mode, ptr(norms), ptr(dout), ctx.scale,
                           stream())
        return (None,) + tuple(_give(p, g) for p, g in zip(params, grads))


_grad_tables = {}


class ToNHWCFn(Function):
    """[N,C,H,W] -> [N,H,W,C] (the decoder_fc -> view(N,64,6,6) seam, autoencoders.py:116-117)."""

    @staticmethod
    def forward(ctx, x):
        x = _check(x, "layout input")
        n, c, h, w = x.shape
        y = torch.empty((n, h, w, c), dtype=torch.float32, device=x.device)
        C.nchw_to_nhwc(ptr(x), ptr(y), n, c, h, w, stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _check(dy, "layout grad")
        n, h, w, c = dy.shape
        dx = torch.empty((n, c, h, w), dtype=torch.float32, device=dy.device)
        C.nhwc_to_nchw(ptr(dy), ptr(dx), n, c, h, w, stream())
        return dx


# ----------------------------------------------------------------------------------------------------------------
# losses — losses/losses.py
# ----------------------------------------------------------------------------------------------------------------
class SqDiffSumFn(Function):
    """sum((a-b)^2) — F.mse_loss(sum) (losses.py:210) — or, with mean=True, sum((a-b)^2) / numel — reconstructionLoss (losses.py:181):
    the fp32 division behind the sum's own rounding, in the reduction's last launch, and (g / numel) formed inside the gradient kernel."""

    @staticmethod
    def forward(ctx, a, b, mean=False):
        a, b = _check(a, "loss input"), _check(b, "loss target")
        assert a.shape == b.shape
        out = torch.empty((), dtype=torch.float32, device=a.device)
        nbytes = C.reduce_workspace(a.numel())
        ws = _ws(nbytes, a.device)
        if mean:
            C.sqdiff_mean(ptr(a), ptr(b), a.numel(), float(a.numel()), ptr(out), ptr(ws), nbytes, stream())
        else:
            C.sqdiff_sum(ptr(a), ptr(b), a.numel(), ptr(out), ptr(ws), nbytes, stream())
        ctx.save_for_backward(a, b)
        ctx.div = float(a.numel()) if mean else None
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        da = db = None
        for k, coef in ((0, 2.0), (1, -2.0)):
            if not ctx.needs_input_grad[k]:
                continue
            d = torch.empty_like(a)
            if ctx.div is None:
                C.sqdiff_grad(ptr(a), ptr(b), ptr(g), coef, ptr(d), a.numel(), stream())
            else:
                C.sqdiff_grad_groups(ptr(a), ptr(b), ptr(g), 0, ctx.div, coef, ptr(d), a.numel(), 1, stream())
            if k == 0:
                da = d
            else:
                db = d
        return da, db, None


def weighted(loss, weight):
    """weight * loss — what every loss function of the reference returns (losses/losses.py:102-256; the trainer itself reads the
    terms from the LossManager) — as one HIP launch: 0 + weight * loss with a separately rounded product is exactly that."""
    return TotalLossFn.apply((float(weight),), None, loss)


def add_scalars(a, b):
    """a + b of two 0-dim loss tensors as one HIP launch (srlz_weighted_total with weights 1, 1: (0 + 1 * a) + 1 * b is exactly a + b)."""
    return TotalLossFn.apply((1.0, 1.0), None, a, b)


class FanOutFn(Function):
    """A tensor with several consumers, made explicit: n aliases go out, and the backward sums the gradients that come back with ONE
    launch in a fixed order (srlz_sum_terms: ((g0 + g1) + g2) + g3) — what autograd would otherwise do with one accumulation kernel
    per extra consumer.  An alias nobody differentiates through contributes nothing."""

    @staticmethod
    def forward(ctx, t, n):
        ctx.set_materialize_grads(False)
        return tuple(t.view_as(t) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        import ctypes
        terms = [_check(g, "fan-out gradient") for g in grads if g is not None]
        if not terms:
            return None, None
        if len(terms) == 1:
            return terms[0], None
        out = torch.empty_like(terms[0])
        while len(terms) > 1:  # (at most four terms per launch)
            head, terms = terms[:4], terms[4:]
            ptrs = (ctypes.c_void_p * len(head))(*[g.data_ptr() for g in head])
            C.sum_terms(ptrs, len(head), ptr(out), out.numel(), stream())
            terms = [out] + terms
            if len(terms) > 1:
                out = torch.empty_like(out)
        return terms[0], None


def fan_out(t, n):
    """n aliases of t whose gradients are summed by one launch (FanOutFn); t itself n times when it carries no gradient."""
    if n <= 1 or not (torch.is_tensor(t) and t.requires_grad and torch.is_grad_enabled()):
        return (t,) * n
    return FanOutFn.apply(t, n)


class Fan(object):
    """Hands out the aliases of a FanOutFn one consumer at a time: Fan(t, n).take() n times.  n <= 1 (or a tensor without gradient):
    the tensor itself."""

    def __init__(self, t, n):
        self.parts = list(fan_out(t, n)) if n > 1 else None
        self.t = t

    def take(self):
        return self.parts.pop() if self.parts else self.t


class CatColsFn(Function):
    """th.cat((a, b), dim=1) of two [B, *] matrices — the input of the inverse / reward heads (forward_inverse.py:62,78-95) — and the
    split of its gradient, one launch each."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _check(a, "cat input"), _check(b, "cat input")
        assert a.dim() == 2 and b.dim() == 2 and a.shape[0] == b.shape[0]
        out = torch.empty((a.shape[0], a.shape[1] + b.shape[1]), dtype=torch.float32, device=a.device)
        C.cat_cols(ptr(a), ptr(b), ptr(out), a.shape[0], a.shape[1], b.shape[1], stream())
        ctx.cols = (a.shape[1], b.shape[1])
        return out

    @staticmethod
    def backward(ctx, d):
        d = _check(d, "cat gradient")
        ca, cb = ctx.cols
        da = torch.empty((d.shape[0], ca), dtype=torch.float32, device=d.device) if ctx.needs_input_grad[0] else None
        db = torch.empty((d.shape[0], cb), dtype=torch.float32, device=d.device) if ctx.needs_input_grad[1] else None
        if da is not None or db is not None:
            C.split_cols(ptr(d), ptr(da), ptr(db), d.shape[0], ca, cb, stream())
        return da, db


class ForwardModelFn(Function):
    """next-state prediction as ONE node (forward_inverse.py:27-37): state + Linear([state ; onehot(action)]).  The concatenation is
    one launch, the residual add sits in the GEMM's epilogue (separately rounded, as `state + self.forward_net(concat)` rounds it);
    backward: d state = dy + (dy . W)[:, :S] — the residual branch's gradient in the data-gradient GEMM's epilogue, only the state
    columns computed — and the weight / bias gradients of the linear layer."""

    @staticmethod
    def forward(ctx, state, action, w, b, n_actions):
        state, w = _check(state, "state"), _check(w, "forward model weight")
        action = action.contiguous()
        assert action.dtype == torch.int64 and action.device == state.device
        m, sdim = state.shape
        k = sdim + n_actions
        assert tuple(w.shape) == (sdim, k)
        cat = torch.empty((m, k), dtype=torch.float32, device=state.device)
        C.concat_onehot(ptr(state), ptr(action), ptr(cat), m, sdim, n_actions, stream())
        y = torch.empty((m, sdim), dtype=torch.float32, device=state.device)
        nbytes = C.linear_workspace(m, sdim, k)
        ws = _ws(nbytes, state.device)
        C.linear_fwd_res(ptr(cat), ptr(w), ptr(b), ptr(state), ptr(y), m, sdim, k, 0, ptr(ws), nbytes, stream())
        ctx.save_for_backward(cat, w)
        ctx.params = (b,)
        ctx.dims = (m, sdim, k)
        return y

    @staticmethod
    def backward(ctx, dy):
        cat, w = ctx.saved_tensors
        dy = _check(dy, "forward model dy")
        m, sdim, k = ctx.dims
        nbytes = C.linear_workspace(m, sdim, k)
        ws = _ws(nbytes, dy.device)
        dstate = None
        if ctx.needs_input_grad[0]:
            dstate = torch.empty((m, sdim), dtype=torch.float32, device=dy.device)
            C.linear_bwd_data_res(ptr(dy), ptr(w), ptr(dy), ptr(dstate), m, sdim, k, sdim, ptr(ws), nbytes, stream())
        dw = _gbuf(w)
        db = _gbuf(ctx.params[0]) if ctx.params[0] is not None else None
        C.linear_bwd_weight(ptr(dy), ptr(cat), ptr(dw), ptr(db), m, sdim, k, ptr(ws), nbytes, stream())
        return dstate, None, _give(w, dw), _give(ctx.params[0], db), None


# ----------------------------------------------------------------------------------------------------------------
# The two frames of a step as ONE batch (models/learner.py:392-393 calls the model on obs, then on next_obs).
# pair_cat joins them (zero-copy when they already are the two halves of one buffer), the model runs once inside
# batch_groups(2), pair_split hands the halves back to the loop body as views.  Losses that see both halves of a pair
# (pair_of) reduce them in one launch and return ONE gradient for the batched tensor.
# ----------------------------------------------------------------------------------------------------------------
def pair_cat(a, b):
    """[a ; b] along dim 0.  No copy when b starts where a ends in the same storage."""
    if is_u8_frames(a) and is_u8_frames(b):  # the loader's bytes: no gradient, any route
        a, b = _check_u8(a, "pair half"), _check_u8(b, "pair half")
        if a.shape == b.shape and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() \
                and b.storage_offset() == a.storage_offset() + a.numel():
            return torch.empty(0, dtype=a.dtype, device=a.device).set_(a.untyped_storage(), a.storage_offset(),
                                                                       (2 * a.shape[0],) + tuple(a.shape[1:]))
        return torch.cat([a, b], 0)
    a, b = frames_as_float(a), frames_as_float(b)
    a, b = _check(a, "pair half"), _check(b, "pair half")
    if a.shape != b.shape:
        raise C.SrlzError("pair_cat: the two halves differ in shape (%s vs %s)" % (tuple(a.shape), tuple(b.shape)))
    shape = (2 * a.shape[0],) + tuple(a.shape[1:])
    same = a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
    if same and b.storage_offset() == a.storage_offset() + a.numel() and not (a.requires_grad or b.requires_grad):
        return torch.empty(0, dtype=a.dtype, device=a.device).set_(a.untyped_storage(), a.storage_offset(), shape)
    return JoinFn.apply(a, b)


class JoinFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        out = torch.empty((2 * a.shape[0],) + tuple(a.shape[1:]), dtype=torch.float32, device=a.device)
        C.join2(ptr(a), ptr(b), ptr(out), a.numel(), stream())
        return out

    @staticmethod
    def backward(ctx, d):
        half = d.shape[0] // 2
        return d[:half], d[half:]


class SplitFn(Function):
    """t [2B, ...] -> (t[:B], t[B:]) as views; the backward joins the two gradients (a missing one counts as zero)."""

    @staticmethod
    def forward(ctx, t):
        half = t.shape[0] // 2
        ctx.half_shape = (half,) + tuple(t.shape[1:])
        ctx.set_materialize_grads(False)
        return t[:half], t[half:]

    @staticmethod
    def backward(ctx, da, db):
        if da is None and db is None:
            return None
        dev = (da if da is not None else db).device
        out = torch.empty((2 * ctx.half_shape[0],) + ctx.half_shape[1:], dtype=torch.float32, device=dev)
        half = out.shape[0] // 2
        if da is not None and db is not None:
            da, db = _check(da, "pair grad"), _check(db, "pair grad")
            if da.numel() % 4 == 0 and da.data_ptr() % 16 == 0 and db.data_ptr() % 16 == 0:
                C.join2(ptr(da), ptr(db), ptr(out), da.numel(), stream())
            else:
                out[:half].copy_(da)
                out[half:].copy_(db)
        else:
            out.zero_()
            (out[:half] if da is not None else out[half:]).copy_(da if da is not None else db)
        return out


def pair_split(t):
    """The two halves of a batched tensor, remembered as a pair so that pair-aware consumers can find the whole."""
    a, b = SplitFn.apply(t)
    # (weak references to the sibling: a <-> b would otherwise be a cycle that keeps the batched tensor — 300 MB for the
    # reconstructions at bs = 256 — alive until Python's cyclic collector runs)
    a._srlz_pair = (t, 0, weakref.ref(b))
    b._srlz_pair = (t, 1, weakref.ref(a))
    return a, b


def pair_of(a, b):
    """The batched tensor whose halves are exactly (a, b) — or None."""
    ra = getattr(a, "_srlz_pair", None)
    if ra is not None and ra[1] == 0 and ra[2]() is b:
        return ra[0]
    if torch.is_tensor(a) and torch.is_tensor(b) and not (a.requires_grad or b.requires_grad) and a.is_cuda and b.is_cuda \
            and a.dtype in (torch.float32, torch.uint8) and a.dtype == b.dtype and a.shape == b.shape \
            and a.is_contiguous() and b.is_contiguous() \
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() \
            and b.storage_offset() == a.storage_offset() + a.numel():
        return pair_cat(a, b)  # two adjacent constant tensors (obs, next_obs): a view over both
    return None


class SqDiffPairLossFn(Function):
    """The reconstruction (mean=True: sum/numel per frame, added) or generation (mean=False: sums added) loss of BOTH frames
    from batched tensors a, b [2B, ...] as ONE scalar: two launches forward, one backward, no scalar glue kernels.  Each
    frame's sum and the combination are rounded exactly like the two-call path's separate fp32 operations."""

    @staticmethod
    def forward(ctx, a, b, mean):
        a, b = _check(a, "loss input"), _check(b, "loss target")
        assert a.shape == b.shape and a.shape[0] % 2 == 0
        sums = torch.empty(2, dtype=torch.float32, device=a.device)
        comb = torch.empty((), dtype=torch.float32, device=a.device)
        nbytes = C.reduce_workspace(a.numel())
        ws = _ws(nbytes, a.device)
        C.sqdiff_pair_loss(ptr(a), ptr(b), a.numel() // 2, 1 if mean else 0, ptr(sums), ptr(comb), ptr(ws), nbytes, stream())
        ctx.save_for_backward(a, b)
        ctx.div = float(a.numel() // 2) if mean else 1.0
        return comb

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _check(g, "loss grad")
        da = db = None
        n = a.numel() // 2
        if ctx.needs_input_grad[0]:
            da = torch.empty_like(a)
            C.sqdiff_grad_groups(ptr(a), ptr(b), ptr(g), 0, ctx.div, 2.0, ptr(da), n, 2, stream())
        if ctx.needs_input_grad[1]:
            db = torch.empty_like(b)
            C.sqdiff_grad_groups(ptr(a), ptr(b), ptr(g), 0, ctx.div, -2.0, ptr(db), n, 2, stream())
        return da, db, None


class KLSumFn(Function):
    """-0.5 * sum(1 + logvar - mu^2 - exp(logvar)) — losses.py:253."""

    @staticmethod
    def forward(ctx, mu, logvar):
        mu, logvar = _check(mu, "mu"), _check(logvar, "logvar")
        out = torch.empty((), dtype=torch.float32, device=mu.device)
        nbytes = C.reduce_workspace(mu.numel())
        ws = _ws(nbytes, mu.device)
        C.kl_sum(ptr(mu), ptr(logvar), mu.numel(), ptr(out), ptr(ws), nbytes, stream())
        ctx.save_for_backward(mu, logvar)
        return out

    @staticmethod
    def backward(ctx, g):
        mu, logvar = ctx.saved_tensors
        g = g.contiguous()
        dmu, dlv = torch.zeros_like(mu), torch.zeros_like(logvar)
        C.kl_grad(ptr(mu), ptr(logvar), ptr(g), 1.0, ptr(dmu), ptr(dlv), mu.numel(), stream())
        return dmu, dlv


class ReparamFn(Function):
    """z = eps * exp(0.5*logvar) + mu — BaseModelVAE.reparameterize, models/models.py:157-163."""

    @staticmethod
    def forward(ctx, mu, logvar, eps):
        mu, logvar, eps = _check(mu, "mu"), _check(logvar, "logvar"), _check(eps, "eps")
        z = torch.empty_like(mu)
        C.reparam_fwd(ptr(mu), ptr(logvar), ptr(eps), ptr(z), mu.numel(), stream())
        ctx.save_for_backward(logvar, eps)
        return z

    @staticmethod
    def backward(ctx, dz):
        logvar, eps = ctx.saved_tensors
        dz = _check(dz, "dz")
        dmu, dlv = torch.empty_like(dz), torch.empty_like(dz)
        C.reparam_bwd(ptr(dz), ptr(logvar), ptr(eps), ptr(dmu), ptr(dlv), dz.numel(), stream())
        return dmu, dlv, None


class CrossEntropyFn(Function):
    """nn.CrossEntropyLoss()(logits, target) (mean) — inverseModelLoss, losses.py:126-127."""

    @staticmethod
    def forward(ctx, logits, target):
        logits = _check(logits, "logits")
        target = target.contiguous()
        assert target.dtype == torch.int64 and target.device == logits.device
        b, a = logits.shape
        out = torch.empty((), dtype=torch.float32, device=logits.device)
        dlogits = torch.empty_like(logits)
        C.cross_entropy(ptr(logits), ptr(target), b, a, ptr(out), ptr(dlogits), stream())
        ctx.save_for_backward(dlogits)
        return out

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        out = torch.empty_like(dlogits)
        C.scale_by_scalar(ptr(dlogits), ptr(_check(g, "loss grad")), 1.0, 1.0, ptr(out), dlogits.numel(), stream())
        return out, None


class PReLUFn(Function):
    """nn.PReLU() with a single slope — EmbeddingNet.fc[0], reference models/triplet.py:24."""

    @staticmethod
    def forward(ctx, x, slope):
        x, slope = _check(x, "prelu input"), _check(slope, "prelu slope")
        assert slope.numel() == 1
        y = torch.empty_like(x)
        C.prelu_fwd(ptr(x), ptr(slope), ptr(y), x.numel(), stream())
        ctx.save_for_backward(x, slope)
        ctx.params = (slope,)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, slope = ctx.saved_tensors
        dy = _check(dy, "prelu dy")
        dx = torch.empty_like(x)
        ds = _gbuf(ctx.params[0])
        C.prelu_bwd(ptr(x), ptr(slope), ptr(dy), ptr(dx), ptr(ds), x.numel(), stream())
        return dx, _give(ctx.params[0], ds)


class TripletLossFn(Function):
    """mean_b relu(|s - p|^2 - |s - n|^2 + alpha) — tripletLoss, reference losses/losses.py:360-376."""

    @staticmethod
    def forward(ctx, s, p, n, alpha):
        s, p, n = _check(s, "anchor states"), _check(p, "positive states"), _check(n, "negative states")
        assert s.shape == p.shape == n.shape and s.dim() == 2
        b, sd = s.shape
        out = torch.empty((), dtype=torch.float32, device=s.device)
        hinge = torch.empty(b, dtype=torch.float32, device=s.device)
        C.triplet_fwd(ptr(s), ptr(p), ptr(n), b, sd, alpha, ptr(out), ptr(hinge), stream())
        ctx.save_for_backward(s, p, n, hinge)
        return out

    @staticmethod
    def backward(ctx, g):
        s, p, n, hinge = ctx.saved_tensors
        g = g.contiguous()
        ds, dp, dn = torch.empty_like(s), torch.empty_like(p), torch.empty_like(n)
        C.triplet_bwd(ptr(s), ptr(p), ptr(n), ptr(hinge), ptr(g), s.shape[0], s.shape[1], ptr(ds), ptr(dp), ptr(dn), stream())
        return ds, dp, dn, None


class ConcatOneHotFn(Function):
    """cat([s, onehot(a)], 1) — forwardModel's input, forward_inverse.py:30 + models/models.py:229-237."""

    @staticmethod
    def forward(ctx, s, a, n_actions):
        s = _check(s, "state")
        a = a.contiguous()
        assert a.dtype == torch.int64 and a.device == s.device
        b, sdim = s.shape
        cat = torch.empty((b, sdim + n_actions), dtype=torch.float32, device=s.device)
        C.concat_onehot(ptr(s), ptr(a), ptr(cat), b, sdim, n_actions, stream())
        ctx.sdim = sdim
        return cat

    @staticmethod
    def backward(ctx, dcat):
        return dcat[:, :ctx.sdim].contiguous(), None, None


def normalize_u8(frames, out=None):
    """uint8 [N,H,W,C] device tensor -> normalised fp32 [N,C,W,H] (the reference's observation tensor); `out`: where to
    write it (a contiguous [N,C,W,H] fp32 tensor, e.g. one half of a pair buffer)."""
    if frames.device.type != "cuda" or frames.dtype != torch.uint8:
        raise C.SrlzError("normalize_u8 expects a uint8 tensor on the GPU")
    frames = frames.contiguous()
    n, h, w, c = frames.shape
    if out is None:
        out = torch.empty((n, c, w, h), dtype=torch.float32, device=frames.device)
    elif tuple(out.shape) != (n, c, w, h) or out.dtype != torch.float32 or not out.is_contiguous():
        raise C.SrlzError("normalize_u8: `out` must be a contiguous float32 [N,C,W,H] tensor")
    C.normalize_u8(ptr(frames), ptr(out), n, h, w, c, stream())
    return out


def fold_grads(grad, stages):
    C.fold_grads(ptr(grad), ptr(stages), grad.numel(), stages.shape[0], stream())


def adam_step(p, g, m, v, lr, step, grad_scale=1.0, betas=(0.9, 0.999), eps=1e-8):
    C.adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, betas[0], betas[1], eps, step, grad_scale, stream())


def adam_step_dev(p, g, m, v, lr, step_dev, bc_dev, grad_scale=1.0, betas=(0.9, 0.999), eps=1e-8):
    C.adam_step_dev(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, betas[0], betas[1], eps, ptr(step_dev), ptr(bc_dev),
                    grad_scale, stream())


def bn_replay(batch_stat, running_mean, running_var):
    C.bn_replay(ptr(batch_stat), BN_MOMENTUM, ptr(running_mean), ptr(running_var), stream())


def bn_replay_many(items):
    """[(batch_stat [128], running_mean, running_var, num_batches_tracked)] of up to 8 layers: one more momentum update each with the
    given batch statistics and the counters += 1, in ONE launch."""
    table = (C.BnReplayItem * len(items))()
    for i, (st, rm, rv, tick) in enumerate(items):
        table[i].batch_stat, table[i].running_mean, table[i].running_var = st.data_ptr(), rm.data_ptr(), rv.data_ptr()
        table[i].num_batches_tracked = tick.data_ptr() if tick is not None else None
    C.bn_replay_many(table, len(items), BN_MOMENTUM, stream())
