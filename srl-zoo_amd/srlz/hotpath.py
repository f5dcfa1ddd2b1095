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


# Test hook that does NOT change the route: when set to a dict, the forwards leave REFERENCES to the tensors the default route produces
# anyway — the pooled maps (whose autograd nodes hold y, the BatchNorm record and the argmax bytes) and the decoder blocks' outputs
# (whose nodes hold the raw input and its BatchNorm record) — so a test can read the ReLU / max-pool decisions of exactly the step the
# product runs (PoolLink, deferred BatchNorm backward, fused block backward, loss in the last ConvTranspose), which TAPS switches off.
OBSERVE = None


def _observe(prefix, idx, t):
    if OBSERVE is not None:
        # (the node's saved tensors are taken NOW: backward frees them)
        OBSERVE["%s.%d" % (prefix, idx)] = (t.detach(), tuple(t.grad_fn.saved_tensors) if t.grad_fn is not None else ())
    return t


def _tap(prefix, idx, t):
    if TAPS is not None and t.requires_grad:
        t.retain_grad()
        TAPS["%s.%d" % (prefix, idx)] = t
    return t


# The fallback routes below are chosen by what the step needs, not by environment variables (those were A/B residue, retired in round 5):
#  * first block as the plain chain Conv1Fn -> BNReLUPoolFn instead of the fused EncInFn: when the image carries a gradient (the
#    perceptual loss back-propagates into the reconstruction) — or when a test clears _FUSE_ENC_IN in-process;
#  * decoder BatchNorm backward materialised instead of deferred into the producing block (ops.BwdLink): when TAPS is on (every
#    d(loss)/dy_k must exist as a tensor) — or when a test clears _DEFER_BN_BWD in-process.
_FUSE_ENC_IN = True
_DEFER_BN_BWD = True


# ---- the step's reconstruction / generation loss taken inside the last ConvTranspose (ops.DecOutLossFn) -------------------------
# SRL4robotics._eagerStep opens `with recon_loss_into(target, mean) as req:` around the batched model call; decoder_forward then
# ends in ONE node that yields the loss scalar (req.loss) and, in place of the reconstruction, the error tensor dec - target
# (the reconstruction itself is not written; SRL4robotics._forwardPair hands None to its caller in place of the decoded frames, so
# nothing downstream can mistake the error for an image).  Steps that need the decoded frames themselves (perceptual loss, callers of
# the model outside _forwardPair's recon request) never open the request; tests clear _FUSE_RECON in-process to compare the two routes.
_FUSE_RECON = True
_RECON = None


class recon_loss_into(object):
    def __init__(self, target, mean):
        self.target, self.mean, self.loss = target, bool(mean), None

    def __enter__(self):
        global _RECON
        self.prev, _RECON = _RECON, (self if _FUSE_RECON else None)
        return self

    def __exit__(self, *exc):
        global _RECON
        _RECON = self.prev
        return False


def _bn_args(bn):
    # nn.BatchNorm2d increments num_batches_tracked once per training-mode call: the counter travels with the running mean
    # (ops._bn_params hands it to srlz_bn_finalize, which advances it by the number of calls the launch stands for)
    bn.running_mean._srlz_tick = bn.num_batches_tracked
    return bn.weight, bn.bias, bn.running_mean, bn.running_var


def encoder_forward(seq, x, training, stat_sink=None, name="encoder_conv"):
    """models/models.py:47-63.  x: [N,C,H,W] (reference layout) -> [N,64,6,6] (NCHW, ready for .view(N,-1))."""
    conv1, bn1, conv2, bn2, conv3, bn3 = seq[0], seq[1], seq[4], seq[5], seq[8], seq[9]
    if not _FUSE_ENC_IN:
        x = ops.frames_as_float(x)  # (only the fused first block reads the loader's bytes)
    # ops.PoolLink: the BatchNorm-backward sums of a pooled block come out of the epilogue of the NEXT convolution's data gradient
    # (which writes d(pooled) anyway); with TAPS on every gradient is looked at from outside: plain chain
    link1 = ops.PoolLink() if TAPS is None else None
    link2 = ops.PoolLink() if TAPS is None else None
    if _FUSE_ENC_IN and not x.requires_grad:  # (an image that carries a gradient needs conv1's data gradient: plain chain)
        p, y = ops.EncInFn.apply(x, conv1.weight, *_bn_args(bn1), training, 1, stat_sink, link1)
        _tap(name, 0, y)
        _tap(name, 3, p)
        _observe(name, 3, p)
    else:
        y, st = ops.Conv1Fn.apply(x, conv1.weight, training)
        _tap(name, 0, y)
        p = _tap(name, 3, ops.BNReLUPoolFn.apply(y, st, *_bn_args(bn1), training, 1, False, stat_sink, link1))
        _observe(name, 3, p)
    y, st = ops.Conv64Fn.apply(p, conv2.weight, None, 1, 1, False, training, link1)
    _tap(name, 4, y)
    p = _observe(name, 7, _tap(name, 7, ops.BNReLUPoolFn.apply(y, st, *_bn_args(bn2), training, 0, False, stat_sink, link2)))
    y, st = ops.Conv64Fn.apply(p, conv3.weight, None, 2, 1, False, training, link2)
    _tap(name, 8, y)
    return _observe(name, 11, _tap(name, 11, ops.BNReLUPoolFn.apply(y, st, *_bn_args(bn3), training, 0, True, stat_sink)))


def replay_encoder_bn(seq, stats, group=0):
    """Second train-mode pass over the same batch (VAE getStates quirk, models/learner.py:402): running statistics
    receive the same batch statistics once more and num_batches_tracked advances; outputs are unchanged.
    `group`: which BatchNorm group's statistics (the batch was one half of a batched pair)."""
    bns = (seq[1], seq[5], seq[9])
    items = [(st[128 * group:128 * (group + 1)], bn.running_mean, bn.running_var, bn.num_batches_tracked) for bn, st in zip(bns, stats)]
    # (one launch for the three layers, counters included — until round 6: three launches + three torch increments per frame)
    ops._ordered_bn_update([bn.running_mean for bn in bns], lambda: ops.bn_replay_many(items))


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
        if TAPS is not None:
            _record_activation(bi + 1, y, st, bn, training)
        out_link = ops.BwdLink() if defer else None
        y, st = ops.DecBlockFn.apply(y, st, *_bn_args(bn), training, conv.weight, conv.bias, training, in_link, out_link)
        in_link = out_link
        _tap("decoder_conv", ci, y)
        _observe("decoder_conv", ci, y)
    bn, last = seq[10], seq[12]
    if TAPS is not None:
        _record_activation(11, y, st, bn, training)
    req = _RECON
    if req is not None and req.loss is None and TAPS is None and torch.is_grad_enabled() and y.shape[0] % 2 == 0 \
            and req.target.shape[0] == y.shape[0] and req.target.shape[1] == last.weight.shape[1] and not req.target.requires_grad:
        req.loss, err = ops.DecOutLossFn.apply(y, st, *_bn_args(bn), training, last.weight, last.bias, in_link, req.target, req.mean)
        _observe("decoder_conv", 12, req.loss)  # (err carries no node: the loss scalar's node holds (y_prev, bnp, w, err))
        return err  # NOT the reconstruction: dec - target (the caller asked for the loss, not for the image)
    return _observe("decoder_conv", 12, _tap("decoder_conv", 12, ops.DecOutFn.apply(y, st, *_bn_args(bn), training, last.weight,
                                                                                      last.bias, in_link)))


def _record_activation(idx, y, st, bn, training):
    """TAPS only: materialise relu(bn(y)) with the same kernel arithmetic the fused loaders use, WITHOUT touching the
    running statistics (a throw-away copy of them is updated instead)."""
    with torch.no_grad():
        rm, rv = bn.running_mean.clone(), bn.running_var.clone()
        bnp, _ = ops._bn_params(st, y.numel() // 64, bn.weight, bn.bias, rm, rv, training, y.device)
        TAPS["decoder_conv.%d" % idx] = ops.bn_relu_materialise(y.detach(), bnp)


# ---------------------------------------------------------------------------------------------------------------------
# Frozen ResNet-18 trunk of EmbeddingNet (reference models/triplet.py:16 -> torchvision resnet18), FORWARD ONLY.
# ---------------------------------------------------------------------------------------------------------------------
def _packed(conv, d):
    """The kernel-layout copy of a frozen convolution's weights, kept on the module and rebuilt only if the parameter was
    moved or written (load_state_dict, .to(device))."""
    key = (conv.weight.data_ptr(), conv.weight._version)
    cached = getattr(conv, "_srlz_pack", None)
    if cached is None or cached[0] != key:
        pk = torch.empty(ops.C.convn_packed_floats(d), dtype=torch.float32, device=conv.weight.device)
        ops.C.convn_pack_weights(ops.ptr(conv.weight), ops.ptr(pk), d, ops.stream())
        cached = (key, pk)
        conv._srlz_pack = cached
    return cached[1]


def _bn_record(bn, stats, tiles, count, training, device, groups=1):
    """One 256-float record per BatchNorm group and block of 64 channels of a C-channel BatchNorm2d, [groups][C / 64][256] (train:
    per-group batch statistics, the running statistics take the groups' momentum updates in order, num_batches_tracked += groups;
    eval: running statistics, the same record for every group).  tiles counts all groups, count is per group."""
    chunks = bn.num_features // 64
    if training:
        bnp = torch.empty(256 * chunks * groups, dtype=torch.float32, device=device)
        nbytes = ops.C.bn_finalize_chunks_workspace(chunks, groups)
        ws = ops._ws(nbytes, device)
        ops.C.bn_finalize_chunks(ops.ptr(stats), tiles, chunks, groups, count, ops.ptr(bn.weight), ops.ptr(bn.bias), ops.BN_EPS,
                                 ops.BN_MOMENTUM, ops.ptr(bn.running_mean), ops.ptr(bn.running_var),
                                 ops.ptr(bn.num_batches_tracked), ops.ptr(bnp), ops.ptr(ws), nbytes, ops.stream())
    else:
        bnp = torch.empty(256 * chunks, dtype=torch.float32, device=device)
        ops.C.bn_eval_params_chunks(ops.ptr(bn.weight), ops.ptr(bn.bias), ops.ptr(bn.running_mean), ops.ptr(bn.running_var),
                                    ops.BN_EPS, chunks, ops.ptr(bnp), ops.stream())
        if groups > 1:
            bnp = bnp.repeat(groups)
    return bnp


def _packed64(conv, d):
    """As _packed, for a 64 -> 64 3x3 convolution in the conv64 kernels' layout (the forward pack only is used)."""
    key = (conv.weight.data_ptr(), conv.weight._version)
    cached = getattr(conv, "_srlz_pack64", None)
    if cached is None or cached[0] != key:
        pk = torch.empty((2, ops.C.conv64_packed_floats()), dtype=torch.float32, device=conv.weight.device)
        ops.C.conv64_pack_weights(ops.ptr(conv.weight), ops.ptr(pk[0]), ops.ptr(pk[1]), d, ops.stream())
        cached = (key, pk)
        conv._srlz_pack64 = cached
    return cached[1]


def _packed_wino(conv):
    """As _packed, for a 64 -> 64 3x3 stride-1 convolution in the Winograd kernel's layout (G g G^T, forward direction)."""
    key = (conv.weight.data_ptr(), conv.weight._version)
    cached = getattr(conv, "_srlz_pack_wino", None)
    if cached is None or cached[0] != key:
        pk = torch.empty(ops.C.conv64_wino_packed_floats(), dtype=torch.float32, device=conv.weight.device)
        ops.C.conv64_wino_pack_weights(ops.ptr(conv.weight), ops.ptr(pk), None, ops.stream())
        cached = (key, pk)
        conv._srlz_pack_wino = cached
    return cached[1]


def _convn(x, conv, bn, training, x_bnp=None, groups=1):
    """raw = conv(x or relu(bn_prev(x))) for an NHWC tensor + the BatchNorm record(s) of `bn` over that output; `groups` independent
    calls batched along n (x_bnp and the returned records are [groups][channel blocks][256])."""
    n, hi, wi, cin = x.shape
    k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    ho, wo = (hi + 2 * p - k) // s + 1, (wi + 2 * p - k) // s + 1
    if cin == 64 and conv.out_channels == 64 and k == 3 and s == 1 and p == 1:
        # layer1 of the trunk (four 64 -> 64 3x3 convolutions at 56 x 56): the auto-encoder's own conv2 kernel — row tables, 16-byte
        # epilogue stores, fused relu(bn(.)) operand — instead of the general-channel-count kernel (0.73 against 0.56 of the matrix peak)
        d = ops.conv64_desc(n, hi, wi, 1, 1, False, groups)
        y = torch.empty((n, ho, wo, 64), dtype=torch.float32, device=x.device)
        if ops.C.conv64_wino_supported(d):
            # ... as Winograd F(2x2, 3x3) (csrc/wino.hip) on even maps: a materialised input (the first convolution of a block) as it is, the
            # raw output of the block's first convolution through the fused relu(bn(.)) landing (x_bnp)
            tiles = ops.C.conv64_wino_tiles(d)
            stats = torch.empty((1, tiles, 128), dtype=torch.float32, device=x.device) if training else None
            ops.C.conv64_wino_fwd(ops.ptr(x), ops.ptr(_packed_wino(conv)), None, ops.ptr(y), ops.ptr(stats), ops.ptr(x_bnp), d, ops.stream())
            return y, _bn_record(bn, stats, tiles, n // groups * ho * wo, training, x.device, groups)
        tiles = ops.C.conv64_fwd_tiles(d)
        stats = torch.empty((1, tiles, 128), dtype=torch.float32, device=x.device) if training else None
        ops.C.conv64_fwd(ops.ptr(x), ops.ptr(_packed64(conv, d)[0]), None, ops.ptr(y), ops.ptr(stats), ops.ptr(x_bnp), d, ops.stream())
        return y, _bn_record(bn, stats, tiles, n // groups * ho * wo, training, x.device, groups)
    d = ops.ConvNDesc(n, hi, wi, ho, wo, cin, conv.out_channels, k, s, p, groups)
    y = torch.empty((n, ho, wo, conv.out_channels), dtype=torch.float32, device=x.device)
    tiles = ops.C.convn_fwd_tiles(d)
    stats = torch.empty((conv.out_channels // 64, tiles, 128), dtype=torch.float32, device=x.device) if training else None
    ops.C.convn_fwd(ops.ptr(x), ops.ptr(_packed(conv, d)), ops.ptr(y), ops.ptr(stats), ops.ptr(x_bnp), d, ops.stream())
    return y, _bn_record(bn, stats, tiles, n // groups * ho * wo, training, x.device, groups)


def resnet18_forward(trunk, x, training, groups=1):
    """torchvision resnet18 up to (and including) avgpool: x [B,3,224,224] (reference layout) -> [B,512] features.
    Forward only (the trunk is frozen, reference models/triplet.py:17-19): runs under no_grad, the result carries no
    gradient.  `training` selects BatchNorm's mode exactly as nn.Module.train()/eval() would.
    groups > 1 (round 6): x holds `groups` trunk calls of B / groups images each, batched along n — the anchor / positive / negative
    views of obs and of next_obs of one time-contrastive step (reference models/learner.py:383-391 + modules.py:92-100: six
    `self.model(view)` calls).  Every BatchNorm works per group (its own batch statistics, the running statistics moved once per group,
    in call order, num_batches_tracked += groups): features, running statistics and counters are those of `groups` separate calls,
    with one launch per layer instead of six and six times the tiles per launch."""
    x = ops.frames_as_float(x)
    require_gpu(x, "resnet18 trunk")
    with torch.no_grad():
        x = ops._check(x, "resnet18 input")
        n, c, h, w = x.shape
        assert c == 3, "the ResNet-18 trunk takes one 3-channel view at a time"
        assert groups >= 1 and n % groups == 0, (n, groups)
        # stem: Conv2d(3,64,7,2,3) -> BatchNorm2d -> ReLU -> MaxPool2d(3,2,1): the kernels of the auto-encoder's first block
        d = ops._skinny_desc(n, c, h, w, 0, groups)
        y = torch.empty((n, d.hf, d.wf, 64), dtype=torch.float32, device=x.device)
        tiles = ops.C.skinny_tiles(d)
        stats = torch.empty((1, tiles, 128), dtype=torch.float32, device=x.device) if training else None
        ops.C.conv1_fwd(ops.ptr(x), ops.ptr(trunk.conv1.weight), ops.ptr(y), ops.ptr(stats), d, ops.stream())
        bnp = _bn_record(trunk.bn1, stats, tiles, n // groups * d.hf * d.wf, training, x.device, groups)
        hp, wp = (d.hf + 2 - 3) // 2 + 1, (d.wf + 2 - 3) // 2 + 1
        act = torch.empty((n, hp, wp, 64), dtype=torch.float32, device=x.device)
        ops.C.bn_relu_pool_fwd(ops.ptr(y), ops.ptr(bnp), ops.ptr(act), None, ops.PoolDesc(n, d.hf, d.wf, hp, wp, 1, 0, groups), ops.stream())
        for layer in (trunk.layer1, trunk.layer2, trunk.layer3, trunk.layer4):
            for block in layer:
                c1, rec1 = _convn(act, block.conv1, block.bn1, training, groups=groups)
                c2, rec2 = _convn(c1, block.conv2, block.bn2, training, x_bnp=rec1, groups=groups)
                out = torch.empty_like(c2)
                pixels, chunks = c2.numel() // c2.shape[3], c2.shape[3] // 64
                if block.downsample is not None:
                    cd, recd = _convn(act, block.downsample[0], block.downsample[1], training, groups=groups)
                    ops.C.bn_add_relu(ops.ptr(c2), ops.ptr(rec2), ops.ptr(cd), ops.ptr(recd), ops.ptr(out), pixels, chunks, groups, ops.stream())
                else:
                    ops.C.bn_add_relu(ops.ptr(c2), ops.ptr(rec2), ops.ptr(act), None, ops.ptr(out), pixels, chunks, groups, ops.stream())
                act = out
        n, ho, wo, ch = act.shape
        feat = torch.empty((n, ch), dtype=torch.float32, device=x.device)
        ops.C.avgpool_nhwc(ops.ptr(act), ops.ptr(feat), n, ho * wo, ch, ops.stream())
    return feat


def linear(layer, x, relu=False):
    return ops.LinearFn.apply(x, layer.weight, layer.bias, relu)


def require_gpu(x, what):
    if not (torch.is_tensor(x) and x.is_cuda):
        raise RuntimeError("%s: the srl-zoo_amd hot path runs on MI355X only (tensor is on %s); there is no CPU "
                           "fallback" % (what, x.device if torch.is_tensor(x) else type(x)))
