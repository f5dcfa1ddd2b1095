"""LossManager and the hot-path losses (reference losses/losses.py:19-59, 102-170, 172-214, 239-256).

Same function names, argument order and return values as the reference; the reductions and their gradients run as
fused HIP kernels (srlz/ops.py).  Losses of other SRL methods (priors, episode/reward priors) are outside the
hot path and not provided.
"""
from __future__ import print_function, division, absolute_import

import torch as th

from srlz import ops


class LossManager:
    """Collects (name, weight, value) triples for one minibatch, forms the weighted total and keeps the history."""

    def __init__(self, model, loss_history=None):
        """
        :param model: (PyTorch model)
        :param loss_history: (dict)
        """
        # trainable, regularisable parameters (biases excluded), as in the reference
        self.reg_params = [param for name, param in model.named_parameters() if
                           ".bias" not in name and param.requires_grad]
        self.loss_history = loss_history
        self.names, self.weights, self.losses = [], [], []
        self.collect_only = False  # (True inside SRL4robotics.trainStep: the loss functions' return values are not formed)

    def addToLosses(self, name, weight, loss_value):
        self.names.append(name)
        self.weights.append(weight)
        self.losses.append(loss_value)

    def lossValues(self):
        """All loss scalars of this minibatch with ONE device->host copy (the reference pays one .item() each)."""
        if not self.losses:
            return []
        return th.stack([l.detach().reshape(()) for l in self.losses]).tolist()

    def updateLossHistory(self, values=None, names=None, weights=None):
        # (names / weights: those of the step the values belong to, when they are booked a step later — learn() reads a step's scalars
        # after it has launched the next one)
        if self.loss_history is not None:
            if values is None:
                values = self.lossValues()
            for name, w, value in zip(self.names if names is None else names, self.weights if weights is None else weights, values):
                if w > 0:
                    if len(self.loss_history[name]) > 0:
                        self.loss_history[name][-1] += w * value
                    else:
                        self.loss_history[name].append(w * value)

    def computeTotalLoss(self):
        return sum([self.weights[i] * self.losses[i] for i in range(len(self.losses))])

    def resetLosses(self):
        self.names, self.weights, self.losses = [], [], []


def _returned(loss, weight, loss_manager):
    """What a loss function returns: weight * loss, as in the reference (losses/losses.py:102-256) — one HIP launch.  The trainer reads
    the terms from the LossManager and ignores the return value; it sets `loss_manager.collect_only` for the duration of its step so
    that the launch (and its autograd node) is not made for nobody."""
    if getattr(loss_manager, "collect_only", False):
        return None
    return ops.weighted(loss, weight)


def l1Loss(params, weight, loss_manager):
    """L1 regularisation: sum over the parameter list of sum(|p|) (reference losses.py:132-142)."""
    l1_loss = ops.ParamNormFn.apply(0, *params)
    loss_manager.addToLosses('l1_loss', weight, l1_loss)
    return _returned(l1_loss, weight, loss_manager)


def l2Loss(params, weight, loss_manager):
    """L2 regularisation: mean over the parameter list of ||p||_2 (reference losses.py:145-155)."""
    l2_loss = ops.ParamNormFn.apply(1, *params)
    loss_manager.addToLosses('l2_loss', weight, l2_loss)
    return _returned(l2_loss, weight, loss_manager)


def rewardModelLoss(rewards_pred, rewards_st, weight, loss_manager):
    """cross-entropy between reward logits and the (categorical) reward (reference losses.py:158-170)."""
    reward_loss = ops.CrossEntropyFn.apply(rewards_pred, rewards_st.view(-1))
    loss_manager.addToLosses('reward_loss', weight, reward_loss)
    return _returned(reward_loss, weight, loss_manager)


def reconstructionLoss(input_image, target_image):
    """sum((a-b)^2) / numel  (reference losses.py:172-181)."""
    return ops.SqDiffSumFn.apply(input_image, target_image, True)


def forwardModelLoss(next_states_pred, next_states, weight, loss_manager):
    """mean squared error between predicted and encoded next states (reference losses.py:102-114)."""
    forward_loss = reconstructionLoss(next_states_pred, next_states)
    loss_manager.addToLosses('forward_loss', weight, forward_loss)
    return _returned(forward_loss, weight, loss_manager)


def inverseModelLoss(actions_pred, actions_st, weight, loss_manager):
    """cross-entropy between action logits and the taken actions (reference losses.py:117-129)."""
    inverse_loss = ops.CrossEntropyFn.apply(actions_pred, actions_st.view(-1))
    loss_manager.addToLosses('inverse_loss', weight, inverse_loss)
    return _returned(inverse_loss, weight, loss_manager)


def _pairSqDiff(a, next_a, b, next_b, mean):
    """sum((a-b)^2)/numel + sum((next_a-next_b)^2)/numel (mean) or the plain sum of the two sums, as ONE op when (a, next_a) and
    (b, next_b) are the halves of batched tensors (the product's batched model call, SRLModules.forwardPair) — every
    intermediate rounded exactly as the per-frame path rounds it; None otherwise."""
    pa, pb = ops.pair_of(a, next_a), ops.pair_of(b, next_b)
    if pa is None or pb is None:
        return None
    return ops.SqDiffPairLossFn.apply(pa, pb, mean)


def autoEncoderLoss(obs, decoded_obs, next_obs, decoded_next_obs, weight, loss_manager):
    """reconstruction error of both frames (reference losses.py:184-196)."""
    ae_loss = _pairSqDiff(obs, next_obs, decoded_obs, decoded_next_obs, True)
    if ae_loss is None:
        ae_loss = ops.add_scalars(reconstructionLoss(obs, decoded_obs), reconstructionLoss(next_obs, decoded_next_obs))
    loss_manager.addToLosses('reconstruction_loss', weight, ae_loss)
    return _returned(ae_loss, weight, loss_manager)


def generationLoss(decoded, next_decoded, obs, next_obs, weight, loss_manager):
    """pixel-wise summed squared error of both frames (reference losses.py:199-214)."""
    generation_loss = _pairSqDiff(decoded, next_decoded, obs, next_obs, False)
    if generation_loss is None:
        generation_loss = ops.add_scalars(ops.SqDiffSumFn.apply(decoded, obs), ops.SqDiffSumFn.apply(next_decoded, next_obs))
    loss_manager.addToLosses('generation_loss', weight, generation_loss)
    return _returned(generation_loss, weight, loss_manager)


def perceptualSimilarityLoss(encoded_real, encoded_prediction, next_encoded_real, next_encoded_prediction,
                             weight, loss_manager):
    """DARLA's perceptual similarity: summed squared distance between the frozen denoiser's encodings of the real frames
    and of the VAE's reconstructions, both frames (reference losses.py:217-236)."""
    pretrained_dae_encoding_loss = ops.add_scalars(ops.SqDiffSumFn.apply(encoded_real, encoded_prediction),
                                                   ops.SqDiffSumFn.apply(next_encoded_real, next_encoded_prediction))
    loss_manager.addToLosses("denoising perceptual similarity", weight, pretrained_dae_encoding_loss)
    return _returned(pretrained_dae_encoding_loss, weight, loss_manager)


def kullbackLeiblerLoss(mu, next_mu, logvar, next_logvar, loss_manager, beta=1):
    """KL(q(z|x) || N(0, I)) summed over elements and batch, both frames (reference losses.py:239-256)."""
    kl_divergence = ops.add_scalars(ops.KLSumFn.apply(mu, logvar), ops.KLSumFn.apply(next_mu, next_logvar))
    loss_manager.addToLosses('kl_loss', beta, kl_divergence)
    return _returned(kl_divergence, beta, loss_manager)


def tripletLoss(states, p_states, n_states, weight, loss_manager, alpha=0.2):
    """Time-contrastive triplet loss: mean relu(|s - p|^2 - |s - n|^2 + alpha) (reference losses.py:360-376)."""
    tcn_triplet_loss = ops.TripletLossFn.apply(states, p_states, n_states, alpha)
    loss_manager.addToLosses('triplet_loss', weight, tcn_triplet_loss)
    return _returned(tcn_triplet_loss, weight, loss_manager)
