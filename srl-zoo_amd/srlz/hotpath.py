"""Composition of the HIP blocks into the reference's two conv stacks.

`encoder_conv` / `decoder_conv` are kept as nn.Sequential containers of standard torch layers (so parameter creation
order, initialisation and state_dict keys are the reference's, models/models.py:47-83), but they are never *called*:
the forward below reads their parameters and runs the MI355X kernels.
"""
import os
import torch

from . import ops

# Debug / test hook: when set to a dict, every intermediate activation of the two conv stacks is recorded under the
# reference's layer name (e.g. "encoder_conv.4" = raw output of conv2) with retain_grad(), so parity tests can compare
# activation gradients layer by layer.  None (default) = no overhead.
TAPS = None


def _tap(prefix, idx, t):
    if TAPS is not None and t.requires_grad:
        t.retain_grad()
        TAPS["%s.%d" % (prefix, idx)] = t
    return t


# A/B switch for the fused first-block backward (ops.EncInFn); the fused form is the product path
_FUSE_ENC_IN = os.environ.get("SRLZ_FUSE_ENC_IN", "1") != "0"


# A/B switch for the deferred decoder BatchNorm backward (ops.BwdLink); deferred is the product path
_DEFER_BN_BWD = os.environ.get("SRLZ_DEFER_BN_BWD", "1") != "0"


def _bn_args(bn):
    return bn.weight, bn.bias, bn.running_mean, bn.running_var


def _tick(bn, training):
    # nn.BatchNorm2d increments num_batches_tracked once per training-mode call (ordered with the other stream's
    # updates of the same layer when the two frames of a step run on two streams)
    if training and bn.num_batches_tracked is not None:
        groups = ops.cur_groups(training)  # a batched pair is `groups` calls of the layer
        ops._ordered_bn_update(bn.running_mean, lambda: bn.num_batches_tracked.add_(groups))


def encoder_forward(seq, x, training, stat_sink=None, name="encoder_conv"):
    """models/models.py:47-63.  x: [N,C,H,W] (reference layout) -> [N,64,6,6] (NCHW, ready for .view(N,-1))."""
    conv1, bn1, conv2, bn2, conv3, bn3 = seq[0], seq[1], seq[4], seq[5], seq[8], seq[9]
    _tick(bn1, training)
    if _FUSE_ENC_IN and not x.requires_grad:  # (an image that carries a gradient needs conv1's data gradient: plain chain)
        p, y = ops.EncInFn.apply(x, conv1.weight, *_bn_args(bn1), training, 1, stat_sink)
        _tap(name, 0, y)
        _tap(name, 3, p)
    else:
        y, st = ops.Conv1Fn.apply(x, conv1.weight, training)
        _tap(name, 0, y)
        p = _tap(name, 3, ops.BNReLUPoolFn.apply(y, st, *_bn_args(bn1), training, 1, False, stat_sink))
    y, st = ops.Conv64Fn.apply(p, conv2.weight, None, 1, 1, False, training)
    _tap(name, 4, y)
    _tick(bn2, training)
    p = _tap(name, 7, ops.BNReLUPoolFn.apply(y, st, *_bn_args(bn2), training, 0, False, stat_sink))
    y, st = ops.Conv64Fn.apply(p, conv3.weight, None, 2, 1, False, training)
    _tap(name, 8, y)
    _tick(bn3, training)
    return _tap(name, 11, ops.BNReLUPoolFn.apply(y, st, *_bn_args(bn3), training, 0, True, stat_sink))


def replay_encoder_bn(seq, stats, group=0):
    """Second train-mode pass over the same batch (VAE getStates quirk, models/learner.py:402): running statistics
    receive the same batch statistics once more and num_batches_tracked advances; outputs are unchanged.
    `group`: which BatchNorm group's statistics (the batch was one half of a batched pair)."""
    for bn, st in zip((seq[1], seq[5], seq[9]), stats):
        def update(bn=bn, st=st[128 * group:128 * (group + 1)]):
            ops.bn_replay(st, bn.running_mean, bn.running_var)
            bn.num_batches_tracked.add_(1)
        ops._ordered_bn_update(bn.running_mean, update)


def decoder_forward(seq, z, training):
    """models/models.py:65-83.  z: [N,64,6,6] NCHW (decoder_fc output viewed) -> [N,C,224,224] NCHW.

    Each BatchNorm-apply + ReLU is fused into the operand load of the transposed convolution that consumes it
    (ops.DecBlockFn / ops.DecOutFn): only the raw convolution outputs y1..y4 exist in memory."""
    a = ops.ToNHWCFn.apply(z)
    y, st = ops.Conv64Fn.apply(a, seq[0].weight, seq[0].bias, 2, 0, True, training)
    _tap("decoder_conv", 0, y)
    # BatchNorm backward deferred into the producing block's kernels (ops.BwdLink); with TAPS on, every d(loss)/dy_k is
    # materialised instead so that retain_grad() on the taps shows true gradients.
    defer = _DEFER_BN_BWD and TAPS is None
    in_link = None  # the first transposed convolution is a plain Conv64Fn: its BatchNorm backward is materialised
    for bi, ci in ((1, 3), (4, 6), (7, 9)):
        bn, conv = seq[bi], seq[ci]
        _tick(bn, training)
        if TAPS is not None:
            _record_activation(bi + 1, y, st, bn, training)
        out_link = ops.BwdLink() if defer else None
        y, st = ops.DecBlockFn.apply(y, st, *_bn_args(bn), training, conv.weight, conv.bias, training, in_link, out_link)
        in_link = out_link
        _tap("decoder_conv", ci, y)
    bn, last = seq[10], seq[12]
    _tick(bn, training)
    if TAPS is not None:
        _record_activation(11, y, st, bn, training)
    return _tap("decoder_conv", 12, ops.DecOutFn.apply(y, st, *_bn_args(bn), training, last.weight, last.bias, in_link))


def _record_activation(idx, y, st, bn, training):
    """TAPS only: materialise relu(bn(y)) with the same kernel arithmetic the fused loaders use, WITHOUT touching the
    running statistics (a throw-away copy of them is updated instead)."""
    with torch.no_grad():
        rm, rv = bn.running_mean.clone(), bn.running_var.clone()
        bnp, _ = ops._bn_params(st, y.numel() // 64, bn.weight, bn.bias, rm, rv, training, y.device)
        TAPS["decoder_conv.%d" % idx] = ops.bn_relu_materialise(y.detach(), bnp)


def linear(layer, x, relu=False):
    return ops.LinearFn.apply(x, layer.weight, layer.bias, relu)


def require_gpu(x, what):
    if not (torch.is_tensor(x) and x.is_cuda):
        raise RuntimeError("%s: the srl-zoo_amd hot path runs on MI355X only (tensor is on %s); there is no CPU "
                           "fallback" % (what, x.device if torch.is_tensor(x) else type(x)))
