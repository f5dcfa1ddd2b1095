"""SRLModules — one encoder chosen by (model_type, losses) + forward / inverse / reward heads
(reference models/modules.py:17-100) and SRLModulesSplit, the split-representation variant (modules.py:103-288).
Only the conv family (`custom_cnn`) is on the MI355X hot path."""
from __future__ import print_function, division, absolute_import

from srlz import ops
from .autoencoders import CNNAutoEncoder
from .vae import CNNVAE
from .triplet import EmbeddingNet
from .forward_inverse import BaseForwardModel, BaseInverseModel, BaseRewardModel
from .models import *  # noqa: F401,F403  (BaseModelSRL, CustomCNN, encodeOneHot, ... as in the reference)



def _forward_pair(module, x, next_x):
    out, next_out = forward_pair(module.forward, x, next_x)
    if isinstance(module.model, BaseModelVAE) and module.model.training:
        # forward -> (decoded, mu, logvar): the learner asks getStates(x), getStates(next_x) next (learner.py:402)
        module.model.rememberPair(x, next_x, out[1], next_out[1])
    return out, next_out


OUT_OF_SCOPE = "model_type '{}' / losses {} are outside the MI355X hot path of this build (custom_cnn with " \
               "autoencoder | vae | dae | inverse | forward | reward | perceptual | triplet); use the reference implementation for them"


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

        if model_type != "custom_cnn":
            raise NotImplementedError(OUT_OF_SCOPE.format(model_type, self.losses))
        if "autoencoder" in self.losses or "dae" in self.losses:
            self.model = CNNAutoEncoder(state_dim)
        elif "vae" in self.losses:
            self.model = CNNVAE(state_dim)
        else:
            self.model = CustomCNN(state_dim)
        if "triplet" in self.losses:
            # (reference modules.py:71-73: whatever was built above is replaced — after it consumed its share of the RNG)
            self.model = EmbeddingNet(state_dim)

    def getStates(self, observations):
        return self.model.getStates(observations)

    def forward(self, x):
        return self.model(x)

    def forwardPair(self, x, next_x):
        """(self(x), self(next_x)) of one training step as a single batched pass (models.forward_pair)."""
        return _forward_pair(self, x, next_x)

    def encode(self, x):
        if "triplet" in self.losses:
            return self.model(x)
        raise NotImplementedError()

    def forwardTriplets(self, anchor, positive, negative):
        """(model(anchor), model(positive), model(negative)) — reference modules.py:92-100."""
        if hasattr(self.model, "forwardViews"):  # one batched pass of the frozen trunk, one BatchNorm group per view
            return tuple(self.model.forwardViews([anchor, positive, negative]))
        return self.model(anchor), self.model(positive), self.model(negative)

    def forwardTripletPair(self, obs, next_obs):
        """The six trunk calls of one time-contrastive step (reference models/learner.py:383-391: forwardTriplets on the three views
        of obs, then of next_obs) as ONE batched pass with six BatchNorm groups in that call order.
        obs / next_obs: [B, 9, W, H] = anchor ; positive ; negative views stacked along channels.
        Returns (states, positive_states, negative_states, next_states) — the reference discards the other two of next_obs."""
        views = [obs[:, :3], obs[:, 3:6], obs[:, 6:], next_obs[:, :3], next_obs[:, 3:6], next_obs[:, 6:]]
        if hasattr(self.model, "forwardViews") and obs.shape == next_obs.shape:
            out = self.model.forwardViews(views, want=[True, True, True, True, False, False])
            return out[0], out[1], out[2], out[3]
        states, positive, negative = self.forwardTriplets(*[v.contiguous() for v in views[:3]])
        next_states, _, _ = self.forwardTriplets(*[v.contiguous() for v in views[3:]])
        return states, positive, negative, next_states


class SRLModulesSplit(BaseForwardModel, BaseInverseModel, BaseRewardModel):
    """Split state representation (reference models/modules.py:103-288): AE / VAE on one block of the state dimensions,
    inverse / forward / reward heads on others; every consumer sees the state with the foreign blocks ZEROED
    (`detachSplit`), so each loss only shapes its own dimensions."""

    def __init__(self, state_dim=2, action_dim=6, cuda=False, model_type="custom_cnn",
                 losses=None, split_dimensions=None, n_hidden_reward=16, inverse_model_type="linear"):
        """
        :param state_dim: (int)
        :param action_dim: (int)
        :param cuda: (bool)
        :param model_type: (str)
        :param losses: ([str])
        :param split_dimensions: (OrderedDict) loss name -> number of dimensions (-1: shared with the previous loss)
        :param n_hidden_reward: (int) hidden units of the reward head
        :param inverse_model_type: (str) 'linear' or 'mlp'
        """
        assert len(split_dimensions) == len(losses), "Please specify as many split dimensions {} as losses {} !". \
            format(len(split_dimensions), len(losses))
        n_dims = sum(split_dimensions.values())
        n_dims += list(split_dimensions.values()).count(-1)  # account for shared dimensions
        assert n_dims == state_dim, \
            "The sum of all splits' dimensions {} must be equal to the state dimension {}" \
            .format(sum(split_dimensions.values()), str(state_dim))

        self.split_dimensions = split_dimensions
        self.model_type = model_type
        self.losses = losses
        BaseForwardModel.__init__(self)
        BaseInverseModel.__init__(self)
        BaseRewardModel.__init__(self)
        self.cuda = cuda
        self.state_dim = state_dim

        self.initForwardNet(self.state_dim, action_dim)
        self.initInverseNet(self.state_dim, action_dim, model_type=inverse_model_type)
        self.initRewardNet(self.state_dim, n_hidden=n_hidden_reward)

        if model_type == "resnet":
            raise ValueError("Resnet not supported when splitting representation")
        if model_type != "custom_cnn":
            raise NotImplementedError(OUT_OF_SCOPE.format(model_type, self.losses))
        if "autoencoder" in losses or "dae" in losses:
            self.model = CNNAutoEncoder(state_dim)
        elif "vae" in losses:
            self.model = CNNVAE(state_dim)
        else:
            self.model = CustomCNN(state_dim)
        if "triplet" in losses:
            raise ValueError("triplet not supported when splitting representation")

    def getStates(self, observations):
        return self.model.getStates(observations)

    def forward(self, x):
        if "autoencoder" in self.losses or "dae" in self.losses:
            return self.forwardAutoencoder(x)
        elif "vae" in self.losses:
            return self.forwardVAE(x)
        return self.model.forward(x)

    def forwardPair(self, x, next_x):
        """(self(x), self(next_x)) of one training step as a single batched pass (models.forward_pair)."""
        return _forward_pair(self, x, next_x)

    def splitRange(self, index):
        """Column range [lo, hi) of the state that `detachSplit(., index)` keeps; (0, 0) when `index` names no split.

        The reference builds the masked state from th.zeros_like blocks and one kept slice (modules.py:191-236); the
        kept slice is the block of `index`, or of the previous sized split when `index` is a shared (-1) one.  A name
        that is not a key keeps nothing: e.g. forwardAutoencoder asks for 'autoencoder' also when the loss is 'dae',
        so a split DAE decodes an all-zero state — reproduced as is."""
        start_idx, pred_dim, kept = 0, 0, (0, 0)
        for key, n_dim in self.split_dimensions.items():
            n_dim = int(n_dim)
            if n_dim == -1:
                if start_idx == 0:
                    raise ValueError("split_dimensions: a shared split (-1) needs a sized split before it")
                if key != index:
                    continue  # shares its dimensions with the previous split
                n_dim = 0
                start_idx -= pred_dim
            if key == index:
                kept = (start_idx, start_idx + (pred_dim if n_dim == 0 else n_dim))
            if n_dim > 0:
                pred_dim = n_dim
                start_idx += n_dim
            else:
                start_idx += pred_dim
        return kept

    def detachSplit(self, tensor, index):
        """`tensor` [B, state_dim] with every split but `index` masked to zero (no gradient flows to them)."""
        lo, hi = self.splitRange(index)
        return ops.MaskColumnsFn.apply(tensor, lo, hi)

    def forwardVAE(self, x):
        input_shape = x.size()
        sink = [] if self.model.training else None
        mu, logvar = self.model.encode(x, stat_sink=sink)
        if self.model.training:
            self.model._remember(x, mu, sink)
        mu_s, logvar_s = self.detachSplit(mu, index='vae'), self.detachSplit(logvar, index='vae')
        z = self.model.reparameterize(mu_s, logvar_s)
        decoded = self.model.decode(z).view(input_shape)
        return decoded, mu_s, logvar_s

    def forwardAutoencoder(self, x):
        input_shape = x.size()
        encoded = self.model.encode(x)
        decoded = self.model.decode(self.detachSplit(encoded, index='autoencoder')).view(input_shape)
        return encoded, decoded

    def inverseModel(self, state, next_state):
        """action logits from the 'inverse' split of [state ; next_state]."""
        return BaseInverseModel.inverseModel(self, self.detachSplit(state, index='inverse'),
                                             self.detachSplit(next_state, index='inverse'))

    def forwardModel(self, state, action):
        """next-state prediction on the 'forward' split."""
        return BaseForwardModel.forwardModel(self, self.detachSplit(state, index='forward'), action)

    def rewardModel(self, state, next_state):
        """reward logits from the 'reward' split of [state ; next_state]."""
        return BaseRewardModel.rewardModel(self, self.detachSplit(state, index='reward'),
                                           self.detachSplit(next_state, index='reward'))
