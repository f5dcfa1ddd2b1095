"""Forward / inverse / reward heads on top of the learned state (reference models/forward_inverse.py:8-95)."""
from __future__ import print_function, division, absolute_import

import torch.nn as nn

from .models import BaseModelSRL
from srlz import hotpath, ops


def _three_layer_head(n_in, n_hidden, n_out):
    """Linear - ReLU - Linear - ReLU - Linear, indices 0 / 2 / 4 as in the reference's state_dict."""
    return nn.Sequential(nn.Linear(n_in, n_hidden), nn.ReLU(), nn.Linear(n_hidden, n_hidden), nn.ReLU(),
                         nn.Linear(n_hidden, n_out))


def _run_head(net, x):
    """A head through the HIP linear kernels (ReLU fused into the first two layers of a 3-layer head)."""
    if isinstance(net, nn.Linear):
        return hotpath.linear(net, x)
    x = hotpath.linear(net[0], x, relu=True)
    x = hotpath.linear(net[2], x, relu=True)
    return hotpath.linear(net[4], x)


class BaseForwardModel(BaseModelSRL):
    def __init__(self):
        self.action_dim = None
        self.forward_net = None
        super(BaseForwardModel, self).__init__()

    def initForwardNet(self, state_dim, action_dim):
        self.action_dim = action_dim
        self.forward_net = nn.Linear(state_dim + action_dim, state_dim)

    def forward(self, x):
        raise NotImplementedError()

    def forwardModel(self, state, action):
        """next-state prediction: state + W [state ; onehot(action)] + b (predicts the delta)."""
        return ops.ForwardModelFn.apply(state, action.view(-1), self.forward_net.weight, self.forward_net.bias, self.action_dim)


class BaseInverseModel(BaseModelSRL):
    def __init__(self):
        self.inverse_net = None
        super(BaseInverseModel, self).__init__()

    def initInverseNet(self, state_dim, action_dim, n_hidden=128, model_type="linear"):
        if model_type not in ("linear", "mlp"):
            raise ValueError("Unknown model_type for inverse model: {}".format(model_type))
        # one Linear, or the 3-layer perceptron; layer creation order = the reference's (same RNG stream, same keys)
        self.inverse_net = nn.Linear(2 * state_dim, action_dim) if model_type == "linear" \
            else _three_layer_head(2 * state_dim, n_hidden, action_dim)

    def forward(self, x):
        raise NotImplementedError()

    def inverseModel(self, state, next_state):
        """action logits from [state ; next_state]."""
        return _run_head(self.inverse_net, ops.CatColsFn.apply(state, next_state))


class BaseRewardModel(BaseModelSRL):
    def __init__(self):
        self.reward_net = None
        super(BaseRewardModel, self).__init__()

    def initRewardNet(self, state_dim, n_rewards=2, n_hidden=16):
        self.reward_net = _three_layer_head(2 * state_dim, n_hidden, n_rewards)

    def forward(self, x):
        raise NotImplementedError()

    def rewardModel(self, state, next_state):
        """reward logits from [state ; next_state] (reference forward_inverse.py:87-95)."""
        return _run_head(self.reward_net, ops.CatColsFn.apply(state, next_state))
