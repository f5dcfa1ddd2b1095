"""CNNAutoEncoder — conv encoder + Linear(2304, S), Linear(S, 2304) + conv decoder (reference models/autoencoders.py:84-118).

The mlp / linear auto-encoders of the reference (autoencoders.py:6-81) are GEMM-only models outside the conv hot
path and are not provided.
"""
from __future__ import print_function, division, absolute_import

import torch.nn as nn

from .models import BaseModelAutoEncoder
from srlz import hotpath


class CNNAutoEncoder(BaseModelAutoEncoder):
    """:param state_dim: (int)"""

    def __init__(self, state_dim=3):
        super(CNNAutoEncoder, self).__init__()
        self.encoder_fc = nn.Sequential(nn.Linear(6 * 6 * 64, state_dim))
        self.decoder_fc = nn.Sequential(nn.Linear(state_dim, 6 * 6 * 64))

    def encode(self, x):
        return hotpath.linear(self.encoder_fc[0], self._encodeConv(x))

    def decode(self, x):
        return self._decodeConv(hotpath.linear(self.decoder_fc[0], x))
