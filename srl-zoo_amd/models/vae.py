"""CNNVAE — conv encoder + two Linear(2304, S) heads (mu, logvar), Linear(S, 2304) + conv decoder
(reference models/vae.py:43-75).  DenseVAE (vae.py:6-40) is an mlp model outside the conv hot path."""
from __future__ import print_function, division, absolute_import

import torch.nn as nn

from .models import BaseModelVAE
from srlz import hotpath


class CNNVAE(BaseModelVAE):
    """:param state_dim: (int)"""

    def __init__(self, state_dim=3):
        super(CNNVAE, self).__init__()
        self.encoder_fc1 = nn.Linear(6 * 6 * 64, state_dim)
        self.encoder_fc2 = nn.Linear(6 * 6 * 64, state_dim)
        self.decoder_fc = nn.Sequential(nn.Linear(state_dim, 6 * 6 * 64))

    def encode(self, x, stat_sink=None):
        e = self._encodeConv(x, stat_sink)
        return hotpath.linear(self.encoder_fc1, e), hotpath.linear(self.encoder_fc2, e)

    def decode(self, z):
        return self._decodeConv(hotpath.linear(self.decoder_fc[0], z))
