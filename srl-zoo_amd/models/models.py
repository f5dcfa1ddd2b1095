"""Network base classes of the image-representation path, MI355X-native.

Public surface = the reference's (models/models.py:16-237): BaseModelSRL, BaseModelAutoEncoder, BaseModelVAE,
CustomCNN, conv3x3, encodeOneHot, with the same method names, return conventions and state_dict keys.  The layer
containers hold the parameters (created by the same torch constructors in the same order, so a seeded construction
reproduces the reference's initial weights bit for bit); `forward` never calls them — it runs the HIP blocks of
srlz/hotpath.py.
"""
from __future__ import print_function, division, absolute_import

import torch as th
import torch.nn as nn

try:
    from preprocessing.preprocess import getNChannels
except ImportError:  # imported as a sub-package from another repository
    from ..preprocessing.preprocess import getNChannels

from srlz import hotpath, ops


def conv3x3(in_planes, out_planes, stride=1):
    """3x3 convolution, padding 1, no bias (reference models/models.py:217-226)."""
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def _encoder_stack():
    # (reference models/models.py:47-63) 224x224xC -> 112 -> pool 56 -> 56 -> pool 27 -> 14 -> pool 6
    spec = [(nn.Conv2d(getNChannels(), 64, kernel_size=7, stride=2, padding=3, bias=False), 1),
            (conv3x3(64, 64, stride=1), 0),
            (conv3x3(64, 64, stride=2), 0)]
    layers = []
    for conv, pool_pad in spec:
        layers += [conv, nn.BatchNorm2d(64), nn.ReLU(inplace=True), nn.MaxPool2d(kernel_size=3, stride=2, padding=pool_pad)]
    return nn.Sequential(*layers)


def _decoder_stack():
    # (reference models/models.py:65-83) 6 -> 13 -> 27 -> 55 -> 111 -> 224xC
    layers = []
    for _ in range(4):
        layers += [nn.ConvTranspose2d(64, 64, kernel_size=3, stride=2), nn.BatchNorm2d(64), nn.ReLU(True)]
    layers.append(nn.ConvTranspose2d(64, getNChannels(), kernel_size=4, stride=2))
    return nn.Sequential(*layers)


class BaseModelSRL(nn.Module):
    """Base class of an SRL network: getStates(observations) retrieves the state (reference models.py:16-33)."""

    def __init__(self):
        super(BaseModelSRL, self).__init__()

    def getStates(self, observations):
        return self.forward(observations)

    def forward(self, x):
        raise NotImplementedError


def forward_pair(forward, x, next_x):
    """(forward(x), forward(next_x)) — the two model calls of a training step (reference models/learner.py:392-393) — as ONE
    batched pass over [x ; next_x] with two BatchNorm groups (srlz.ops.batch_groups): half the kernel launches, twice the
    work per launch, per-call BatchNorm semantics intact.  Tensors of the returned tuples are halves of batched tensors."""
    xx = ops.pair_cat(x, next_x)
    with ops.batch_groups(2):
        out = forward(xx)
    if isinstance(out, tuple):
        halves = [ops.pair_split(t) for t in out]
        return tuple(h[0] for h in halves), tuple(h[1] for h in halves)
    return ops.pair_split(out)


class BaseModelAutoEncoder(BaseModelSRL):
    """Auto-encoder family: owns the conv encoder and decoder stacks (reference models.py:36-114)."""

    def __init__(self):
        super(BaseModelAutoEncoder, self).__init__()
        self.encoder_conv = _encoder_stack()
        self.decoder_conv = _decoder_stack()

    # -- HIP hot path ---------------------------------------------------------------------------------------------
    def _encodeConv(self, x, stat_sink=None):
        """x [N,C,224,224] -> flattened [N, 2304] in the reference's NCHW order (index c*36 + h*6 + w)."""
        hotpath.require_gpu(x, "encoder")
        e = hotpath.encoder_forward(self.encoder_conv, x, self.training, stat_sink)
        return e.view(e.size(0), -1)

    def _decodeConv(self, z):
        """z [N, 2304] (NCHW order) -> [N,C,224,224]."""
        return hotpath.decoder_forward(self.decoder_conv, z.view(z.size(0), 64, 6, 6), self.training)

    # -- reference surface ----------------------------------------------------------------------------------------
    def getStates(self, observations):
        return self.encode(observations)

    def encode(self, x):
        raise NotImplementedError

    def decode(self, x):
        raise NotImplementedError

    def forward(self, x):
        input_shape = x.size()
        # the state feeds the decoder AND whatever the caller does with it (forward / inverse / reward heads): an explicit fan-out,
        # whose backward sums the returning gradients in one launch (ops.FanOutFn)
        encoded, to_decoder = ops.fan_out(self.encode(x), 2)
        decoded = self.decode(to_decoder).view(input_shape)
        return encoded, decoded


class BaseModelVAE(BaseModelAutoEncoder):
    """VAE family (reference models.py:117-176): forward -> (decoded, mu, logvar); getStates -> mu."""

    def __init__(self):
        super(BaseModelVAE, self).__init__()
        # (input tensor, mu, [encoder BN batch statistics]) of the most recent training-mode forwards
        self._recent = []
        self._recent_versions = []
        # optional override of the noise source: callable(mu) -> eps tensor (parity tests feed the oracle's eps)
        self.eps_fn = None

    def getStates(self, observations):
        # The reference's learner calls getStates(obs) right after forward(obs) in TRAIN mode (learner.py:402):
        # a second, identical encoder pass whose only effects are mu (bit-identical) and one more running-stat
        # update with the same batch statistics.  Reproduce exactly that without recomputing the pass.
        if self.training:
            for i, (x_ref, mu, stats, group) in enumerate(self._recent):
                if x_ref is observations and x_ref._version == self._recent_versions[i]:
                    hotpath.replay_encoder_bn(self.encoder_conv, stats, group)
                    return mu
        return self.encode(observations)[0]

    def _remember(self, x, mu, stats, group=0):
        if len(self._recent) >= 2:
            self._recent.pop(0)
            self._recent_versions.pop(0)
        self._recent.append((x, mu, stats, group))
        self._recent_versions.append(x._version)

    def forgetRecent(self):
        """Drop the remembered forwards: after an optimiser step their mu is stale (a later getStates must re-encode, as
        the reference would), and they pin two input batches."""
        self._recent, self._recent_versions = [], []

    def rememberPair(self, x, next_x, mu=None, next_mu=None):
        """After a batched forward over [x ; next_x]: make the learner's getStates(x) / getStates(next_x) (the quirk above)
        find the halves — group 0 / group 1 of the statistics the batched pass recorded.  What getStates returns is
        encode()[0], the FULL mu (reference models.py:131-139): the halves are taken from the batched mu that the forward
        remembered, never from what a wrapper returned as its second output (SRLModulesSplit.forwardVAE hands out the mu
        masked to the 'vae' split).  `mu` / `next_mu` are re-used only when they are the halves of exactly that tensor."""
        _, mu_full, stats, _ = self._recent.pop()
        self._recent_versions.pop()
        self._recent, self._recent_versions = [], []
        pair = getattr(mu, "_srlz_pair", None) if mu is not None else None
        if pair is None or pair[0] is not mu_full or next_mu is None:
            mu, next_mu = ops.pair_split(mu_full)
        self._remember(x, mu, stats, 0)
        self._remember(next_x, next_mu, stats, 1)

    def reparameterize(self, mu, logvar):
        """z = eps * exp(0.5 logvar) + mu in training (eps from torch's generator, as the reference does), mu in eval."""
        if self.training:
            if self.eps_fn is None:
                # one draw per MODEL CALL, in call order (reference models.py:161 runs once per self.model(...)): a batched pair
                # consumes the generator exactly like the two calls it stands for
                eps = th.empty_like(mu)
                for part in eps.chunk(ops.cur_groups(True)):
                    part.normal_()
            elif ops.cur_groups(True) > 1:  # (test hook) a batched pair: one draw per model call, in call order
                eps = th.cat([self.eps_fn(part) for part in mu.chunk(ops.cur_groups(True))], 0)
            else:
                eps = self.eps_fn(mu)
            return ops.ReparamFn.apply(mu, logvar, eps)
        return mu

    def forward(self, x):
        input_shape = x.size()
        sink = [] if self.training else None
        mu, logvar = self.encode(x, stat_sink=sink)
        # mu: sampled from, returned (the KL term) and remembered for getStates (the heads); logvar: sampled from and returned —
        # explicit fan-outs (ops.FanOutFn sums the returning gradients in one launch)
        mu, mu_z, mu_states = ops.fan_out(mu, 3)
        logvar, logvar_z = ops.fan_out(logvar, 2)
        if self.training:
            self._remember(x, mu_states, sink)
        z = self.reparameterize(mu_z, logvar_z)
        decoded = self.decode(z).view(input_shape)
        return decoded, mu, logvar


class CustomCNN(BaseModelSRL):
    """Conv encoder + Linear(2304, state_dim) (reference models.py:179-214)."""

    def __init__(self, state_dim=2):
        super(CustomCNN, self).__init__()
        self.conv_layers = _encoder_stack()
        self.fc = nn.Linear(6 * 6 * 64, state_dim)

    def forward(self, x):
        hotpath.require_gpu(x, "CustomCNN")
        e = hotpath.encoder_forward(self.conv_layers, x, self.training, None)
        return hotpath.linear(self.fc, e.view(e.size(0), -1))


def encodeOneHot(tensor, n_dim):
    """One-hot encoding of an int64 column tensor (reference models.py:229-237)."""
    encoded_tensor = th.zeros(tensor.shape[0], n_dim, device=tensor.device)
    return encoded_tensor.scatter_(1, tensor.data, 1.)
