"""SRLModules — one encoder chosen by (model_type, losses) + forward / inverse / reward heads
(reference models/modules.py:17-100).  Only the conv family (`custom_cnn`) is on the MI355X hot path."""
from __future__ import print_function, division, absolute_import

from .autoencoders import CNNAutoEncoder
from .vae import CNNVAE
from .forward_inverse import BaseForwardModel, BaseInverseModel, BaseRewardModel
from .models import *  # noqa: F401,F403  (BaseModelSRL, CustomCNN, encodeOneHot, ... as in the reference)

OUT_OF_SCOPE = "model_type '{}' / losses {} are outside the MI355X hot path of this build (custom_cnn with " \
               "autoencoder | vae | dae | inverse | forward); use the reference implementation for them"


class SRLModules(BaseForwardModel, BaseInverseModel, BaseRewardModel):
    def __init__(self, state_dim=2, action_dim=6, cuda=False, model_type="custom_cnn", losses=None,
                 inverse_model_type="linear"):
        """
        :param state_dim: (int)
        :param action_dim: (int)
        :param cuda: (bool)
        :param model_type: (str)
        :param losses: ([str])
        :param inverse_model_type: (str) 'linear' or 'mlp'
        """
        self.model_type = model_type
        self.losses = losses if losses is not None else []
        BaseForwardModel.__init__(self)
        BaseInverseModel.__init__(self)
        BaseRewardModel.__init__(self)
        self.cuda = cuda  # (sic) the reference stores the flag under this name, shadowing nn.Module.cuda()

        # creation order == RNG consumption order == state_dict order of the reference (modules.py:37-73)
        self.initForwardNet(state_dim, action_dim)
        self.initInverseNet(state_dim, action_dim, model_type=inverse_model_type)
        self.initRewardNet(state_dim)

        if model_type != "custom_cnn" or "triplet" in self.losses:
            raise NotImplementedError(OUT_OF_SCOPE.format(model_type, self.losses))
        if "autoencoder" in self.losses or "dae" in self.losses:
            self.model = CNNAutoEncoder(state_dim)
        elif "vae" in self.losses:
            self.model = CNNVAE(state_dim)
        else:
            self.model = CustomCNN(state_dim)

    def getStates(self, observations):
        return self.model.getStates(observations)

    def forward(self, x):
        return self.model(x)

    def encode(self, x):
        raise NotImplementedError()

    def forwardTriplets(self, anchor, positive, negative):
        raise NotImplementedError(OUT_OF_SCOPE.format(self.model_type, ["triplet"]))
