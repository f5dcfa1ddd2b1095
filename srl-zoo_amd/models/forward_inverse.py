"""Forward / inverse / reward heads on top of the learned state (reference models/forward_inverse.py:8-95)."""
from __future__ import print_function, division, absolute_import

import torch as th
import torch.nn as nn

from .models import BaseModelSRL
from srlz import hotpath, ops


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
        concat = ops.ConcatOneHotFn.apply(state, action.view(-1), self.action_dim)
        return state + hotpath.linear(self.forward_net, concat)


class BaseInverseModel(BaseModelSRL):
    def __init__(self):
        self.inverse_net = None
        super(BaseInverseModel, self).__init__()

    def initInverseNet(self, state_dim, action_dim, n_hidden=128, model_type="linear"):
        if model_type == "linear":
            self.inverse_net = nn.Linear(state_dim * 2, action_dim)
        elif model_type == "mlp":
            self.inverse_net = nn.Sequential(nn.Linear(state_dim * 2, n_hidden), nn.ReLU(),
                                             nn.Linear(n_hidden, n_hidden), nn.ReLU(),
                                             nn.Linear(n_hidden, action_dim))
        else:
            raise ValueError("Unknown model_type for inverse model: {}".format(model_type))

    def forward(self, x):
        raise NotImplementedError()

    def inverseModel(self, state, next_state):
        """action logits from [state ; next_state]."""
        x = th.cat((state, next_state), dim=1)
        if isinstance(self.inverse_net, nn.Linear):
            return hotpath.linear(self.inverse_net, x)
        x = hotpath.linear(self.inverse_net[0], x, relu=True)
        x = hotpath.linear(self.inverse_net[2], x, relu=True)
        return hotpath.linear(self.inverse_net[4], x)


class BaseRewardModel(BaseModelSRL):
    def __init__(self):
        self.reward_net = None
        super(BaseRewardModel, self).__init__()

    def initRewardNet(self, state_dim, n_rewards=2, n_hidden=16):
        self.reward_net = nn.Sequential(nn.Linear(2 * state_dim, n_hidden), nn.ReLU(),
                                        nn.Linear(n_hidden, n_hidden), nn.ReLU(),
                                        nn.Linear(n_hidden, n_rewards))

    def forward(self, x):
        raise NotImplementedError()

    def rewardModel(self, state, next_state):
        """reward logits from [state ; next_state] (reference forward_inverse.py:87-95)."""
        x = th.cat((state, next_state), dim=1)
        x = hotpath.linear(self.reward_net[0], x, relu=True)
        x = hotpath.linear(self.reward_net[2], x, relu=True)
        return hotpath.linear(self.reward_net[4], x)
