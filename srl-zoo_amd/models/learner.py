"""BaseLearner / SRL4robotics — the trainer plugin surface of the reference (models/learner.py:40-579), MI355X-native.

Constructor arguments, module-level knobs (N_EPOCHS, BATCH_SIZE, ... set by train.py), ``learn()``'s signature and
return triple, the files written and the exit codes are the reference's.  What differs is underneath:
  * the model's forward/backward, the losses and Adam run as HIP kernels (srlz/);
  * all parameters live in one flat buffer; one fused Adam step; with torch.distributed initialised (one process per
    GPU, backend "nccl" == RCCL) ONE all-reduce of the flat gradient bucket per step (srlz/optim.py);
  * loss scalars are read back once per step (one D2H copy) instead of once per loss.
There is no CPU path: ``cuda=False`` (or no GPU) raises.
"""
from __future__ import print_function, division, absolute_import

import json
import os
import sys
import time
from collections import defaultdict, OrderedDict
from pprint import pprint

import numpy as np
import torch as th

from losses.losses import LossManager, autoEncoderLoss, forwardModelLoss, inverseModelLoss, kullbackLeiblerLoss, \
    generationLoss, rewardModelLoss, l1Loss, l2Loss, perceptualSimilarityLoss, tripletLoss
from pipeline import NAN_ERROR
from preprocessing.data_loader import DataLoader
from utils import printRed, detachToNumpy, printYellow
from srlz import optim
from .modules import SRLModules, SRLModulesSplit

MAX_BATCH_SIZE_GPU = 256  # minibatch size used when predicting states
EPOCH_FLAG = 1            # print every epoch
N_WORKERS = 4

# set by train.py from the command line (reference train.py:72-77)
DISPLAY_PLOTS = True
BATCH_SIZE = 256
N_EPOCHS = 1
VALIDATION_SIZE = 0.2
BALANCED_SAMPLING = False
# build-specific: ship decoded frames as uint8 and normalise on the GPU (bit-identical, 4x less PCIe traffic)
RAW_UINT8_INPUT = True
# build-specific: decode every frame ONCE — frames that arrived are kept in HBM and later epochs gather their minibatches by index
# (preprocessing/resident.py); False restores the reference's re-decoding of every epoch (A/B, tests)
RESIDENT_FRAMES = True

SUPPORTED_LOSSES = {"autoencoder", "vae", "dae", "forward", "inverse", "reward", "perceptual", "random", "triplet"}


def n_epochs_planned(losses):
    """Epochs learn() will run (reference learner.py:346-350: none for the 'random' features)."""
    return 0 if (len(losses) == 1 and losses[0] == 'random') else N_EPOCHS


def _requireGpu(cuda):
    if not cuda or not th.cuda.is_available():
        raise RuntimeError("srl-zoo_amd runs the training hot path on MI355X only: cuda=%s, torch.cuda.is_available()=%s. "
                           "There is deliberately no CPU fallback (use the reference implementation on CPU)."
                           % (cuda, th.cuda.is_available()))


class _DeviceFeed(object):
    """One-minibatch look-ahead between the loader process and the GPU: the H2D copy of minibatch i+1 is issued on a copy
    stream right after step i has been enqueued (advance()), i.e. BEFORE the host blocks on step i's loss scalars, so it
    travels while step i computes.  Iteration order and contents are the loader's."""

    def __init__(self, loader, device):
        self.it, self.device = iter(loader), device
        self.copy = th.cuda.Stream(device=device)
        self.ahead = None
        self.exhausted = False  # the loader's end-of-epoch marker was seen: never pull from it again (the loader process
        self.advance()          # is already producing the NEXT epoch, which belongs to the next _DeviceFeed)

    def _load(self):
        try:
            item = next(self.it)
        except StopIteration:
            self.exhausted = True
            return None
        with th.cuda.stream(self.copy):
            # (idx, obs, next_obs, ...): the two frames land as the halves of ONE device buffer, which is how the batched
            # model call and the pair-aware losses want them (BaseLearner._toDevicePair then has nothing to copy)
            pair = len(item) >= 3 and th.is_tensor(item[1]) and th.is_tensor(item[2]) and item[1].shape == item[2].shape \
                and item[1].dtype == item[2].dtype and item[1].dim() == 4
            moved = [t.to(self.device, non_blocking=True) if th.is_tensor(t) and not (pair and i in (1, 2)) else t
                     for i, t in enumerate(item)]
            if pair:
                n = item[1].shape[0]
                both = th.empty((2 * n,) + tuple(item[1].shape[1:]), dtype=item[1].dtype, device=self.device)
                both[:n].copy_(item[1], non_blocking=True)
                both[n:].copy_(item[2], non_blocking=True)
                moved[1], moved[2] = both[:n], both[n:]
            moved = tuple(moved)
        done = th.cuda.Event()
        done.record(self.copy)
        return moved, done

    def advance(self):
        """Start moving the next minibatch (no-op if one is already on its way or the epoch is over)."""
        if self.ahead is None and not self.exhausted:
            self.ahead = self._load()

    def __iter__(self):
        return self

    def __next__(self):
        self.advance()
        if self.ahead is None:
            raise StopIteration
        (item, done), self.ahead = self.ahead, None
        cur = th.cuda.current_stream(self.device)
        cur.wait_event(done)
        for t in item:
            if th.is_tensor(t):
                t.record_stream(cur)
        return item


class BaseLearner(object):
    """Base class of a state-representation learner.

    :param state_dim: (int)
    :param batch_size: (int)
    :param seed: (int)
    :param cuda: (bool)
    """

    def __init__(self, state_dim, batch_size, seed=1, cuda=False):
        super(BaseLearner, self).__init__()
        self.state_dim = state_dim
        self.batch_size = batch_size
        self.model = None
        self.seed = seed
        self.use_dae = False
        np.random.seed(seed)
        th.manual_seed(seed)
        if cuda and th.cuda.is_available():
            th.cuda.manual_seed(seed)
        self.device = th.device("cuda" if th.cuda.is_available() and cuda else "cpu")

    # Layout of the uint8 frames handed to _toDevice / _toDevicePair: "planar" = [B, C, W, H] (DataLoader(raw_uint8="planar"), what
    # learn() builds its loaders with), "nhwc" = [B, H, W, C] (DataLoader(raw_uint8=True): the frames as decoded).  Stated by whoever
    # builds the loader — never guessed from a shape (a 3 x 3 x 6 x 9 frame fits both).
    frame_layout = "planar"

    def _isPlanar(self, frames):
        """uint8 frames in the reference's tensor layout [B, C, W, H] (see frame_layout)."""
        if frames.dtype != th.uint8:
            return False
        if self.frame_layout not in ("planar", "nhwc"):
            raise ValueError("frame_layout must be 'planar' or 'nhwc', got %r" % (self.frame_layout,))
        return self.frame_layout == "planar"

    def _readsBytes(self):
        """Whether the training step can take the loader's bytes as they are (subclasses that own such a step say so)."""
        return False

    def _toDevice(self, frames):
        """Loader minibatch -> observation tensor on the device.  uint8 frames (DataLoader(raw_uint8=...)) are
        normalised (and, for [B, H, W, C] frames, laid out) by the GPU; float tensors are the reference's ready-made observations."""
        frames = frames.to(self.device, non_blocking=True)
        if frames.dtype == th.uint8:
            from srlz import ops
            frames = ops.frames_as_float(frames) if self._isPlanar(frames) else ops.normalize_u8(frames)
        return frames

    def _toDevicePair(self, frames, next_frames):
        """Both frames of a minibatch -> (obs, next_obs) that are the two HALVES OF ONE device buffer, so that the batched
        model call (SRLModules.forwardPair) and the pair-aware reconstruction loss need no concatenation copy."""
        from srlz import ops
        frames = frames.to(self.device, non_blocking=True)
        next_frames = next_frames.to(self.device, non_blocking=True)
        if frames.shape != next_frames.shape:
            return self._toDevice(frames), self._toDevice(next_frames)
        n = frames.shape[0]
        if self._isPlanar(frames):
            # the reference's layout already, still bytes: a step whose only readers of the observations are conv1 and the fused
            # reconstruction loss takes them as they are (srlz_conv1_fwd_u8 ...); anything else gets the float tensor
            keep = self._readsBytes()
            if keep and frames.is_contiguous() and next_frames.is_contiguous() \
                    and frames.untyped_storage().data_ptr() == next_frames.untyped_storage().data_ptr() \
                    and next_frames.storage_offset() == frames.storage_offset() + frames.numel():
                return frames, next_frames  # (the feed delivered them as one buffer already)
            both = th.empty((2 * n,) + tuple(frames.shape[1:]), dtype=th.uint8 if keep else th.float32, device=self.device)
            if keep:
                both[:n].copy_(frames, non_blocking=True)
                both[n:].copy_(next_frames, non_blocking=True)
            else:
                ops.frames_as_float(frames, out=both[:n])
                ops.frames_as_float(next_frames, out=both[n:])
        elif frames.dtype == th.uint8:
            _, h, w, c = frames.shape
            both = th.empty((2 * n, c, w, h), dtype=th.float32, device=self.device)
            ops.normalize_u8(frames, out=both[:n])
            ops.normalize_u8(next_frames, out=both[n:])
        else:
            both = ops.pair_cat(frames, next_frames)
        return both[:n], both[n:]

    def _predFn(self, observations):
        """Observations -> states (np.ndarray), model in whatever mode the caller set."""
        return detachToNumpy(self.model.getStates(observations))

    def predStatesWithDataLoader(self, data_loader):
        """States for every minibatch the loader yields, concatenated."""
        predictions = []
        for obs_var in data_loader:
            if obs_var.shape[0] == 0:  # the test minibatch list may end with an empty range
                continue
            predictions.append(self._predFn(self._toDevice(obs_var)))
        return np.concatenate(predictions, axis=0)

    def learn(self, *args, **kwargs):
        raise NotImplementedError("Learn method not implemented")

    @staticmethod
    def saveStates(states, images_path, rewards, log_folder, name=""):
        """image_to_state<name>.json and states_rewards<name>.npz (reference learner.py:97-118)."""
        print("Saving image path to state representation (image_to_state{}.json)".format(name))
        image_to_state = {path: list(map(str, state)) for path, state in zip(images_path, states)}
        with open("{}/image_to_state{}.json".format(log_folder, name), 'w') as f:
            json.dump(image_to_state, f, sort_keys=True)
        print("Saving states and rewards (states_rewards{}.npz)".format(name))
        np.savez('{}/states_rewards{}.npz'.format(log_folder, name), states=states, rewards=rewards)


class SRL4robotics(BaseLearner):
    """Trainer for the conv auto-encoder / VAE / forward-inverse family.

    Arguments as in the reference (models/learner.py:121-148).  `split_dimensions` (OrderedDict with a positive sum)
    selects SRLModulesSplit; `l1_reg` / `l2_reg` > 0 add the regularisers; `path_to_dae` / `state_dim_dae` name the
    pre-trained denoiser of the perceptual loss (`--losses vae perceptual`).
    """

    def __init__(self, state_dim, model_type="resnet", inverse_model_type="linear", log_folder="logs/default",
                 seed=1, learning_rate=0.001, l1_reg=0.0, l2_reg=0.0, cuda=False,
                 multi_view=False, losses=None, losses_weights_dict=None, n_actions=6, beta=1,
                 split_dimensions=-1, path_to_dae=None, state_dim_dae=200, occlusion_percentage=None):
        super(SRL4robotics, self).__init__(state_dim, BATCH_SIZE, seed, cuda)
        losses = list(losses) if losses is not None else []
        unsupported = set(losses) - SUPPORTED_LOSSES
        if unsupported:
            raise NotImplementedError("losses %s are outside the MI355X hot path of this build (supported: %s)"
                                      % (sorted(unsupported), sorted(SUPPORTED_LOSSES)))

        self.multi_view = multi_view
        self.losses = losses
        self.dim_action = n_actions
        self.beta = beta
        self.use_forward_loss = "forward" in losses
        self.use_inverse_loss = "inverse" in losses
        self.use_reward_loss = "reward" in losses
        self.use_autoencoder = "autoencoder" in losses
        self.use_vae = "vae" in losses
        self.use_dae = "dae" in losses
        self.use_triplets = "triplet" in losses
        if self.use_triplets and (self.use_vae or self.use_autoencoder or self.use_dae):
            # the reference itself crashes on `vae triplet` (mu / logvar are never bound, learner.py:383-414 vs 457-459) and
            # silently drops the auto-encoder for the EmbeddingNet: there is no behaviour to reproduce
            raise NotImplementedError("'triplet' replaces the model by EmbeddingNet (reference modules.py:71-73): it cannot be "
                                      "combined with autoencoder / vae / dae")
        self.perceptual_similarity_loss = "perceptual" in losses
        self.path_to_dae = path_to_dae
        self.denoiser = None

        if isinstance(split_dimensions, OrderedDict) and sum(split_dimensions.values()) > 0:
            printYellow("Using splitted representation")
            self.model = SRLModulesSplit(state_dim=self.state_dim, action_dim=self.dim_action, model_type=model_type,
                                         cuda=cuda, losses=losses, split_dimensions=split_dimensions,
                                         inverse_model_type=inverse_model_type)
        else:
            self.model = SRLModules(state_dim=self.state_dim, action_dim=self.dim_action, model_type=model_type,
                                    cuda=cuda, losses=losses, inverse_model_type=inverse_model_type)
        print("Using {} model".format(model_type))

        _requireGpu(cuda)
        self.cuda = cuda
        self.device = th.device("cuda", th.cuda.current_device())
        self.model = self.model.to(self.device)
        self.rank, self.world_size = optim.world()

        # SRLZ_GRAPH=1: hipGraph replay of the step body (see _graphStep).  Measured on MI355X: 3 % at bs = 32, nothing at
        # bs >= 64 — the small kernels of a step cost ~10 us each ON THE GPU whether they are enqueued one by one or replayed
        # from a graph, so the cure for small minibatches is fewer kernels (which is what the batched pair delivers).
        # The two frames of a step run as ONE batched model call with two BatchNorm groups (SRLModules.forwardPair): half the
        # launches, twice the grid of every small layer, per-call BatchNorm semantics intact.
        self._use_pair = True  # (False: two model calls per step — frames of different shape take that route; tests set it)
        self._use_graph = os.environ.get("SRLZ_GRAPH", "0") == "1"
        self._graphs = {}
        from srlz import ops as _ops
        _ops.norm_lut(self.device)  # built eagerly: its first use must not fall inside a stream capture or on a side stream

        # one flat parameter / gradient buffer + fused Adam (torch.optim.Adam defaults)
        self.flat_params = optim.FlatParams(self.model)
        self.optimizer = optim.FusedAdam(self.flat_params, lr=learning_rate)
        self.log_folder = log_folder
        self.model_type = model_type

        self.losses_weights_dict = {"forward": 1.0, "inverse": 2.0, "reward": 1.0, "priors": 1.0,
                                    "episode-prior": 1.0, "reward-prior": 10, "triplet": 1.0,
                                    "autoencoder": 1.0, "vae": 0.5e-6, "perceptual": 1e-6, "dae": 1.0,
                                    'l1_reg': l1_reg, "l2_reg": l2_reg, 'random': 1.0}
        self.occlusion_percentage = occlusion_percentage
        self.state_dim_dae = state_dim_dae
        if losses_weights_dict is not None:
            self.losses_weights_dict.update(losses_weights_dict)
        if self.use_dae and self.occlusion_percentage is not None:
            print("Using a maximum occlusion surface of {}".format(str(self.occlusion_percentage)))

    @staticmethod
    def loadSavedModel(log_folder, valid_models, cuda=True):
        """Rebuild a learner from <log_folder>/exp_config.json + srl_model.pth.
        :return: (SRL4robotics, OrderedDict)"""
        assert os.path.exists(log_folder), "Error: folder '{}' does not exist".format(log_folder)
        assert os.path.exists(log_folder + "exp_config.json"), \
            "Error: could not find 'exp_config.json' in '{}'".format(log_folder)
        assert os.path.exists(log_folder + "srl_model.pth"), \
            "Error: could not find 'srl_model.pth' in '{}'".format(log_folder)
        with open(log_folder + 'exp_config.json', 'r') as f:
            exp_config = json.load(f, object_pairs_hook=OrderedDict)  # keep the order of the losses

        losses = exp_config['losses']
        difference = set(losses).symmetric_difference(valid_models)
        assert set(losses).intersection(valid_models) != set(), "Error: Not supported losses " + ", ".join(difference)

        if exp_config.get('multi-view', False):
            import preprocessing.preprocess as pre
            pre.N_CHANNELS = 6
        srl_model = SRL4robotics(exp_config['state-dim'], model_type=exp_config['model-type'], cuda=cuda,
                                 multi_view=exp_config.get('multi-view', False), losses=losses,
                                 n_actions=exp_config['n_actions'],
                                 split_dimensions=exp_config.get('split-dimensions', -1),
                                 inverse_model_type=exp_config.get('inverse-model-type', 'linear'),
                                 occlusion_percentage=exp_config.get('occlusion-percentage', 0))
        srl_model.model.load_state_dict(th.load(log_folder + 'srl_model.pth', map_location=srl_model.device))
        return srl_model, exp_config

    # ---------------------------------------------------------------------------------------------------------
    def loadDenoiser(self, state_dict):
        """The pre-trained, frozen DAE of the perceptual loss (reference learner.py:317-326): eval mode, no gradients to
        its parameters — only to the images it encodes."""
        self.denoiser = SRLModules(state_dim=self.state_dim_dae, action_dim=self.dim_action, model_type="custom_cnn",
                                   cuda=self.cuda, losses=["dae"])
        self.denoiser.load_state_dict(state_dict)
        self.denoiser.eval()
        self.denoiser = self.denoiser.to(self.device)
        for param in self.denoiser.parameters():
            param.requires_grad = False

    def saveModel(self, path):
        """th.save(state_dict) with the reference's keys and NCHW shapes (CPU tensors, loadable anywhere).  With several GPUs
        every rank calls this: the BatchNorm running statistics are averaged over the ranks first (a collective), rank 0
        writes the file."""
        sd = optim.average_running_stats(OrderedDict((k, v.detach().clone()) for k, v in self.model.state_dict().items()))
        if self.rank == 0:
            th.save(OrderedDict((k, v.cpu()) for k, v in sd.items()), path)

    def _readsBytes(self):
        """True when the ONLY readers of the step's observations are the first convolution (forward and weight gradient) and the
        reconstruction / generation loss inside the last ConvTranspose — the kernels that take the loader's uint8 frames as they
        are (ops.EncInFn / ops.DecOutLossFn): the default AE / VAE steps and the heads-only steps (inverse / forward / reward on the
        CustomCNN encoder) of the custom_cnn models.  Everything else
        (DAE noise, perceptual loss, triplets, the ResNet trunks, graph replay, the A/B switches that undo those fusions) gets the
        normalised float tensor (ops.frames_as_float)."""
        from srlz import hotpath
        return (RAW_UINT8_INPUT and self.model_type == "custom_cnn" and self._use_pair
                and not self._use_graph and not self.use_triplets and not self.use_dae
                and not (self.use_vae and self.perceptual_similarity_loss)
                and hotpath._FUSE_RECON and hotpath._FUSE_ENC_IN and hotpath.TAPS is None)

    def _forwardPair(self, x, next_x, recon=None):
        """(self.model(x), self.model(next_x)) — in that program order for the BatchNorm running statistics — as ONE batched
        model call with two BatchNorm groups when possible.

        recon = (target, next_target, mean): the caller only needs the reconstruction / generation LOSS of the two decoded
        frames against these targets, not the frames: the loss is then taken inside the last ConvTranspose (ops.DecOutLossFn)
        and returned as third value (None when that was not possible: the caller computes the loss from the decoded frames).
        (Until round 5 one launch took at most 1149 images — the pooling kernels' grid.y — and minibatches above 574 samples fell back
        to two model calls; the grids are one-dimensional now and what bounds a call is memory: ~80 MB of activations per image.)"""
        from srlz import hotpath, ops
        if self._use_pair and x.shape == next_x.shape and not self.use_triplets:
            target = ops.pair_of(recon[0], recon[1]) if recon is not None else None
            if target is not None:
                with hotpath.recon_loss_into(target, recon[2]) as req:
                    out = self.model.forwardPair(x, next_x)
                if req.loss is not None:
                    # the decoder's last kernel kept the loss and stored dec - target in place of the reconstruction: the
                    # "decoded frames" of the two outputs are NOT images — hand None to the caller so that a consumer fails loudly
                    # (auto-encoders return (states, decoded), VAEs (decoded, mu, logvar): reference models.py:102-106,174-182)
                    at = 0 if self.use_vae else 1
                    out = tuple(tuple(None if j == at else t for j, t in enumerate(o)) for o in out)
                return out[0], out[1], req.loss
            return self.model.forwardPair(x, next_x) + ((None,) if recon is not None else ())
        if recon is not None:
            return self._forwardPair(x, next_x) + (None,)
        return self.model(x), self.model(next_x)

    def trainStep(self, obs, next_obs, actions_st, loss_manager, validation_mode=False, noisy_obs=None,
                  next_noisy_obs=None, rewards_st=None):
        """The minibatch-loop body of the reference (models/learner.py:362-497) on device tensors.

        Runs forward, losses, backward (also on validation minibatches, as the reference does), the gradient
        all-reduce and the Adam step.  Returns the total loss as a 0-dim device tensor; per-loss tensors stay in
        `loss_manager`.  With hipGraph mode on (SRLZ_GRAPH, see _graphStep) the same body is replayed from a captured
        graph: one launch per step instead of ~300.
        """
        if self._use_graph and self.world_size == 1:
            return self._graphStep(loss_manager, validation_mode, dict(obs=obs, next_obs=next_obs, actions_st=actions_st,
                                                                       noisy_obs=noisy_obs, next_noisy_obs=next_noisy_obs,
                                                                       rewards_st=rewards_st))
        return self._eagerStep(obs, next_obs, actions_st, loss_manager, validation_mode, noisy_obs, next_noisy_obs,
                               rewards_st)

    # -- hipGraph mode ----------------------------------------------------------------------------------------------
    # The step body can be captured ONCE per variant (training / validation) into a HIP graph
    # — forward, losses, backward, gradient delivery, Adam with a device-side step counter — and replayed on static input
    # buffers.  Capturing needs warm-up executions of the body (allocator, lazy initialisation); the parameters, Adam
    # moments, BatchNorm buffers and the RNG state they disturb are snapshotted and restored, so a graphed run follows the
    # eager one step for step.  Single GPU only (the RCCL all-reduce stays outside graphs in this build).
    def _stateTensors(self):
        tensors = [self.flat_params.flat, self.optimizer.m, self.optimizer.v, self.optimizer.t_dev]
        tensors += [b for b in self.model.buffers()]
        return tensors

    def _graphStep(self, loss_manager, validation_mode, inputs):
        entry = self._graphs.get(validation_mode)
        if entry is None:
            self.optimizer.use_device_step()
            static = {k: (th.empty_like(v) if v is not None else None) for k, v in inputs.items()}
            for k, v in inputs.items():
                if v is not None:
                    static[k].copy_(v)
            snapshot = [t.clone() for t in self._stateTensors()]
            rng = th.cuda.get_rng_state(self.device)
            host_t = self.optimizer.t
            warm_lm = LossManager(self.model, None)
            cur = th.cuda.current_stream(self.device)
            side = th.cuda.Stream(device=self.device)
            side.wait_stream(cur)
            with th.cuda.stream(side):
                for _ in range(2):
                    self._eagerStep(loss_manager=warm_lm, validation_mode=validation_mode, **static)
            cur.wait_stream(side)
            th.cuda.synchronize(self.device)
            for t, saved in zip(self._stateTensors(), snapshot):
                t.copy_(saved)
            th.cuda.set_rng_state(rng, self.device)
            self.optimizer.t = host_t
            cap_lm = LossManager(self.model, None)
            graph = th.cuda.CUDAGraph()
            with th.cuda.graph(graph):
                loss = self._eagerStep(loss_manager=cap_lm, validation_mode=validation_mode, **static)
            self.optimizer.t = host_t
            entry = dict(graph=graph, static=static, loss=loss, lm=cap_lm)
            self._graphs[validation_mode] = entry
        for k, v in inputs.items():
            if v is not None:
                entry["static"][k].copy_(v)
        if validation_mode:
            self.model.eval()
        else:
            self.model.train()
            self.optimizer.t += 1
        entry["graph"].replay()
        loss_manager.names, loss_manager.weights = list(entry["lm"].names), list(entry["lm"].weights)
        loss_manager.losses = list(entry["lm"].losses)
        self._last_obs = inputs["obs"]
        return entry["loss"]

    def _eagerStep(self, obs, next_obs, actions_st, loss_manager, validation_mode=False, noisy_obs=None,
                   next_noisy_obs=None, rewards_st=None):
        if validation_mode:
            self.model.eval()
        else:
            self.model.train()
        self.optimizer.zero_grad()
        loss_manager.resetLosses()

        from srlz import ops
        if ops.is_u8_frames(obs) and not self._readsBytes():
            obs, next_obs = ops.frames_as_float(obs), ops.frames_as_float(next_obs)
        decoded_obs = decoded_next_obs = None
        recon_loss = None  # the reconstruction / generation loss when it was taken inside the last ConvTranspose
        if self.use_triplets:
            # anchor / positive / negative views stacked along channels (reference learner.py:383-391: six trunk calls per step, each
            # with its own BatchNorm batch statistics) — here ONE batched pass of the frozen trunk with six BatchNorm groups in the
            # reference's call order (round 6; `_use_pair = False`: the six separate passes, kept for the tests)
            if self._use_pair:
                states, positive_states, negative_states, next_states = self.model.forwardTripletPair(obs, next_obs)
            else:
                states, positive_states, negative_states = self.model.model(obs[:, :3].contiguous()), \
                    self.model.model(obs[:, 3:6].contiguous()), self.model.model(obs[:, 6:].contiguous())
                next_states = self.model.model(next_obs[:, :3].contiguous())
                self.model.model(next_obs[:, 3:6].contiguous())
                self.model.model(next_obs[:, 6:].contiguous())
        elif self.use_autoencoder:
            (states, decoded_obs), (next_states, decoded_next_obs), recon_loss = self._forwardPair(obs, next_obs, (obs, next_obs, True))
        elif self.use_dae:
            (states, decoded_obs), (next_states, decoded_next_obs), recon_loss = self._forwardPair(noisy_obs, next_noisy_obs,
                                                                                                  (obs, next_obs, True))
        elif self.use_vae and not self.perceptual_similarity_loss:
            (decoded_obs, mu, logvar), (decoded_next_obs, next_mu, next_logvar), recon_loss = self._forwardPair(
                obs, next_obs, (obs, next_obs, False))
            states, next_states = self.model.getStates(obs), self.model.getStates(next_obs)
        elif self.use_vae:
            (decoded_obs, mu, logvar), (decoded_next_obs, next_mu, next_logvar) = self._forwardPair(obs, next_obs)
            states, next_states = self.model.getStates(obs), self.model.getStates(next_obs)
            if self.perceptual_similarity_loss:
                # the frozen denoiser's encodings of the real frames and of the reconstructions (reference
                # learner.py:404-412; only the states of its (states, decoded) pairs are used, so its decoder is skipped)
                states_denoiser = self.denoiser.getStates(obs)
                next_states_denoiser = self.denoiser.getStates(next_obs)
                states_denoiser_predicted = self.denoiser.getStates(decoded_obs)
                next_states_denoiser_predicted = self.denoiser.getStates(decoded_next_obs)
        else:
            states, next_states = self._forwardPair(obs, next_obs)

        w = self.losses_weights_dict
        # the states feed several heads (and, as next_states, the forward loss): explicit fan-outs, one alias per consumer, so that the
        # gradients coming back are summed by one launch each (ops.FanOutFn) instead of one accumulation kernel per extra consumer
        n_heads = int(self.use_forward_loss) + int(self.use_inverse_loss) + int(self.use_reward_loss)
        states_fan, next_states_fan = ops.Fan(states, n_heads + int(self.use_triplets)), ops.Fan(next_states, n_heads)
        # same order as the reference's loop body (learner.py:420-449): regularisers, forward, inverse, reward, AE, VAE
        # (the terms are read from the LossManager below; the loss functions' own `weight * loss` return values would be one launch
        # per term for nobody)
        loss_manager.collect_only = True
        if w['l1_reg'] > 0:
            l1Loss(loss_manager.reg_params, w['l1_reg'], loss_manager)
        if w['l2_reg'] > 0:
            l2Loss(loss_manager.reg_params, w['l2_reg'], loss_manager)
        if self.use_forward_loss:
            next_states_pred = self.model.forwardModel(states_fan.take(), actions_st)
            forwardModelLoss(next_states_pred, next_states_fan.take(), weight=w['forward'], loss_manager=loss_manager)
        if self.use_inverse_loss:
            actions_pred = self.model.inverseModel(states_fan.take(), next_states_fan.take())
            inverseModelLoss(actions_pred, actions_st, weight=w['inverse'], loss_manager=loss_manager)
        if self.use_reward_loss:
            rewards_pred = self.model.rewardModel(states_fan.take(), next_states_fan.take())
            rewardModelLoss(rewards_pred, rewards_st, weight=w['reward'], loss_manager=loss_manager)
        if (self.use_autoencoder or self.use_dae) and recon_loss is not None:
            # (decoded_* are None here: the loss came out of the decoder's last kernel and the reconstruction was never written)
            loss_manager.addToLosses('reconstruction_loss', w["dae" if self.use_dae else "autoencoder"], recon_loss)
        elif self.use_autoencoder or self.use_dae:
            autoEncoderLoss(ops.frames_as_float(obs), decoded_obs, ops.frames_as_float(next_obs), decoded_next_obs,
                            weight=w["dae" if self.use_dae else "autoencoder"], loss_manager=loss_manager)
        if self.use_vae:
            kullbackLeiblerLoss(mu, next_mu, logvar, next_logvar, loss_manager=loss_manager, beta=self.beta)
            if self.perceptual_similarity_loss:
                perceptualSimilarityLoss(states_denoiser, states_denoiser_predicted, next_states_denoiser,
                                         next_states_denoiser_predicted, weight=w['perceptual'],
                                         loss_manager=loss_manager)
            elif recon_loss is not None:
                loss_manager.addToLosses('generation_loss', w['vae'], recon_loss)
            else:
                generationLoss(decoded_obs, decoded_next_obs, ops.frames_as_float(obs), ops.frames_as_float(next_obs),
                               weight=w['vae'], loss_manager=loss_manager)

        if self.use_triplets:
            tripletLoss(states_fan.take(), positive_states, negative_states, weight=w['triplet'], loss_manager=loss_manager, alpha=0.2)

        loss_manager.collect_only = False
        # LossManager.computeTotalLoss() as one launch, which also drops the step's scalars [total, l_0, l_1, ...] into the tail
        # of the gradient bucket: with several GPUs every rank reads the SAME (mean) losses back, so the NaN exit and the
        # best-model decision are taken by all ranks together
        if len(loss_manager.losses) >= self.flat_params.TAIL:
            # (the step's scalars — total first — ride in the TAIL slots behind the gradients, whichever branch runs below)
            raise ValueError("a training step carries at most {} loss terms (got {}: {})".format(
                self.flat_params.TAIL - 1, len(loss_manager.losses), loss_manager.names))
        if len(loss_manager.losses) >= 1:
            loss = ops.TotalLossFn.apply(tuple(loss_manager.weights), self.flat_params.tail, *loss_manager.losses)
            loss.backward()  # the reference backpropagates on validation minibatches too (learner.py:487-489)
        else:
            loss = loss_manager.computeTotalLoss()
            loss.backward()
            self.flat_params.put_scalars([loss] + list(loss_manager.losses))
        if not validation_mode:
            grad_scale = optim.allreduce_gradients(self.flat_params)
            self.optimizer.step(grad_scale)
        else:
            self.flat_params.discard()
            optim.allreduce_scalars(self.flat_params)
        if hasattr(self.model.model, "forgetRecent"):
            self.model.model.forgetRecent()  # (VAE) the cached mu of this step's forwards is stale from here on
        self._last_obs = obs
        return loss

    def learn(self, images_path, actions, rewards, episode_starts):
        """
        Learn a state representation.
        :param images_path: (numpy 1D array)
        :param actions: (np.ndarray)
        :param rewards: (numpy 1D array)
        :param episode_starts: (numpy 1D array) True where an episode starts
        :return: (loss_history dict, learned states np.ndarray [N, state_dim], [(loss name, weight)])
        """
        print("\nYour are using the following weights for the losses:")
        pprint(self.losses_weights_dict)

        # ---- minibatches: every index whose successor belongs to the same episode, shuffled, full batches, sorted
        num_samples = images_path.shape[0] - 1
        indices = np.array([i for i in range(num_samples) if not episode_starts[i + 1]], dtype='int64')
        np.random.shuffle(indices)
        minibatchlist = [np.array(sorted(indices[start_idx:start_idx + self.batch_size]))
                         for start_idx in range(0, len(indices) - self.batch_size + 1, self.batch_size)]
        test_minibatchlist = DataLoader.createTestMinibatchList(len(images_path), MAX_BATCH_SIZE_GPU)

        n_val_batches = np.round(VALIDATION_SIZE * len(minibatchlist)).astype(np.int64)
        val_indices = np.random.permutation(len(minibatchlist))[:n_val_batches]
        print("{} minibatches for training, {} samples".format(len(minibatchlist) - n_val_batches,
                                                               (len(minibatchlist) - n_val_batches) * BATCH_SIZE))
        print("{} minibatches for validation, {} samples".format(n_val_batches, n_val_batches * BATCH_SIZE))
        assert n_val_batches > 0, "Not enough sample to create a validation set"
        if self.world_size > 1:  # minibatches are sharded train / validation separately, ragged tails dropped
            assert n_val_batches // self.world_size > 0 and (len(minibatchlist) - n_val_batches) // self.world_size > 0, \
                "Not enough minibatches for {} GPUs: every rank needs at least one training and one validation " \
                "minibatch per epoch ({} / {} available)".format(self.world_size, len(minibatchlist) - n_val_batches,
                                                                n_val_batches)

        n_actions = int(np.max(actions) + 1)
        # the cross-entropy / one-hot kernels index by target: reject what nn.CrossEntropyLoss / scatter_ would reject
        if (self.use_inverse_loss or self.use_forward_loss) and (n_actions > self.dim_action or int(np.min(actions)) < 0):
            raise ValueError("actions must lie in [0, {}) (n_actions of the model), found [{}, {}]".format(
                self.dim_action, int(np.min(actions)), int(np.max(actions))))
        if self.use_reward_loss:
            # validated AFTER the reference's own transformation (learner.py:444-449: -1 -> 0, then .long() truncates, so
            # fractional rewards such as 0.5 / -0.5 are legal there): the class index must lie in [0, 2)
            classes = np.array(rewards).copy()
            classes[classes == -1] = 0
            classes = th.from_numpy(np.ascontiguousarray(classes)).long().numpy()
            if classes.size and (classes.min() < 0 or classes.max() > 1):
                raise ValueError("the reward head has two classes: after mapping -1 to 0 and truncating to int64 (reference "
                                 "learner.py:444-449) rewards must fall in [0, 2), found classes {}".format(
                                     sorted(set(classes.tolist()))))
        print("{} unique actions / {} actions".format(len(set(actions)), n_actions))
        print("Number of observations per action")
        print(np.array([np.sum(actions == i) for i in range(n_actions)], dtype=np.int64))

        if self.use_vae and self.perceptual_similarity_loss and self.path_to_dae is not None:
            self.loadDenoiser(th.load(self.path_to_dae, map_location=self.device))

        # The decoded dataset stays resident (uint8 [frames, C, W, H], in HBM when it fits the budget): epoch 1 streams from the
        # loader process as in the reference and is absorbed, later epochs receive INDICES from that same process (same per-epoch
        # permutation from the same forked RNG, same end-of-epoch marker) and gather on the device.  Triplets too: the negative view of a
        # frame is camera 1 of ANOTHER time step of the same record (reference data_loader.py:219-243) — an index into the same store,
        # which keeps the two views of every time step; the loader process keeps drawing it with the reference's random.randint.
        use_bytes = bool(RAW_UINT8_INPUT)
        resident = fill = None
        # Host cores (several ranks share one machine): decoding threads x loader processes of this rank x local ranks <= usable
        # cores, and the loader processes pinned to the cores next to this rank's GPU (the reference's N_WORKERS = 4 assumes it owns
        # the host: data_loader.py:129-193 is one process of one trainer)
        will_fill = self.world_size > 1 and RESIDENT_FRAMES and use_bytes and n_epochs_planned(self.losses) > 1
        n_workers = optim.loader_workers(N_WORKERS, passes=2 if will_fill else 1) if self.world_size > 1 else N_WORKERS
        affinity = optim.numa_cpus_of_device(self.device.index) if (self.world_size > 1 and self.device.type == "cuda") else None
        self.loader_placement = {"n_workers": n_workers, "requested": N_WORKERS, "cpu_affinity": affinity,
                                 "usable_cores": optim.usable_cores()}
        if RESIDENT_FRAMES and use_bytes and n_epochs_planned(self.losses) > 1 \
                and (not self.use_triplets or DataLoader.negativesIndexable(images_path)):
            from preprocessing.resident import ResidentFrames
            import preprocessing.preprocess as _pre
            frame_shape = (6 if self.use_triplets else _pre.getNChannels(), _pre.IMAGE_WIDTH, _pre.IMAGE_HEIGHT)
            # (the DAE's device-side occlusion reads the store in HBM: no store at all when it would not fit there)
            # (... decided by ALL ranks together: free HBM differs between ranks, and a store on some ranks only would pair their
            # slice exchange at the epoch boundary with the others' gradient all-reduce)
            if optim.all_ranks(not self.use_dae or ResidentFrames.fits_device(len(images_path), frame_shape, self.device)):
                needed = np.concatenate([np.concatenate((mb, mb + 1)) for mb in minibatchlist])
                resident = ResidentFrames(len(images_path), frame_shape, self.device, needed, rank=self.rank,
                                          world_size=self.world_size)
            if resident is not None and self.world_size > 1:
                # several ranks: the epoch-1 stream of a rank carries a random 1/W of the minibatches, so the store is NOT completed
                # by absorbing it — every rank decodes its own fixed slice of the dataset beside the first epoch (a second loader
                # process, in order, bytes only) and the slices are exchanged at the epoch boundary (ResidentFrames.exchange)
                from preprocessing.resident import FillPass
                fill = FillPass(resident, images_path, n_workers=n_workers, multi_view=self.multi_view, cpu_affinity=affinity)
        self._resident = resident
        data_loader = DataLoader(minibatchlist, images_path, n_workers=n_workers, multi_view=self.multi_view,
                                 use_triplets=self.use_triplets, is_training=True, apply_occlusion=self.use_dae,
                                 occlusion_percentage=self.occlusion_percentage, rank=self.rank,
                                 world_size=self.world_size, val_indices=val_indices, cpu_affinity=affinity,
                                 raw_uint8="planar" if use_bytes and (not self.use_dae or resident is not None) else False,
                                 index_switch=resident is not None)
        def makeTestLoader(minibatches=test_minibatchlist):
            # (forked when it is needed — at the end of learn(): with the dataset resident the states are predicted from the store)
            return DataLoader(minibatches, images_path, n_workers=n_workers, multi_view=self.multi_view, use_triplets=self.use_triplets,
                              cpu_affinity=affinity,
                              max_queue_len=1, is_training=False, apply_occlusion=self.use_dae,
                              occlusion_percentage=self.occlusion_percentage,
                              raw_uint8="planar" if use_bytes and not self.use_dae else False,
                              infinite_loop=minibatches is test_minibatchlist)

        loss_history = defaultdict(list)
        loss_manager = LossManager(self.model, loss_history)
        best_error = np.inf
        best_model_path = "{}/srl_model.pth".format(self.log_folder)
        start_time = time.time()

        n_epochs = N_EPOCHS
        if len(self.losses) == 1 and self.losses[0] == 'random':
            n_epochs = 0
            printYellow("Skipping training because using random features")
            self.saveModel(best_model_path)

        val_set = set(int(i) for i in val_indices)
        # build-specific record (train.py writes it to <log_folder>/epoch_stats.json): wall seconds, frames and index-only minibatches
        # of every epoch — what shows epoch 1 decode-bound and the later epochs GPU-bound
        self.epoch_stats = []
        for epoch in range(n_epochs):
            epoch_loss, epoch_batches, val_loss, val_batches = 0.0, 0, 0.0, 0
            pending = None  # the previous step's scalars: (ticket, validation?, loss names, loss weights)
            epoch_t0, epoch_gathers = time.time(), (resident.gathers if resident is not None else 0)
            feed = _DeviceFeed(data_loader, self.device)
            for minibatch_num, (minibatch_idx, obs, next_obs, noisy_obs, next_noisy_obs) in enumerate(feed):
                validation_mode = int(minibatch_idx) in val_set
                if obs is None:
                    # an index-only minibatch: its frames are resident; gather [obs ; next_obs] (and the DAE's occluded copies, from
                    # the rectangles the loader process drew) on the device
                    mb = minibatchlist[minibatch_idx]
                    if self.use_triplets:  # (the loader process drew the negatives: frame indices in the two trailing slots)
                        obs, next_obs = resident.triplet_pair(mb, noisy_obs, next_noisy_obs)
                        noisy_obs = next_noisy_obs = None
                    else:
                        if self.use_dae:
                            noisy_obs, next_noisy_obs = resident.occluded_pair(mb, noisy_obs, next_noisy_obs)
                        obs, next_obs = resident.pair(mb)
                    obs, next_obs = self._toDevicePair(obs, next_obs)
                else:
                    if self.use_dae:
                        noisy_obs, next_noisy_obs = self._toDevicePair(noisy_obs, next_noisy_obs)
                    if resident is not None and obs.dtype == th.uint8:
                        obs, next_obs = obs.to(self.device, non_blocking=True), next_obs.to(self.device, non_blocking=True)
                        if self.world_size == 1 and resident.absorb(minibatchlist[minibatch_idx], obs, next_obs) \
                                and not data_loader.index_mode.is_set():
                            data_loader.shipIndices()  # every frame a minibatch can ask for is home: no more pixels, no more decoding
                    obs, next_obs = self._toDevicePair(obs, next_obs)
                actions_st = th.from_numpy(actions[minibatchlist[minibatch_idx]]).view(-1, 1).to(self.device)

                rewards_st = None
                if self.use_reward_loss:
                    rewards_st = rewards[minibatchlist[minibatch_idx]].copy()
                    rewards_st[rewards_st == -1] = 0  # removing negative reward (reference learner.py:439-441)
                    rewards_st = th.from_numpy(rewards_st).to(self.device).long()

                loss = self.trainStep(obs, next_obs, actions_st, loss_manager, validation_mode, noisy_obs,
                                      next_noisy_obs, rewards_st)
                feed.advance()  # next minibatch's H2D copy overlaps this step (issued before the host waits below)
                if fill is not None:
                    fill.drain()  # whatever chunks of the own slice are decoded by now
                # one D2H copy for every scalar of this step (total first; the mean over the ranks when there are several) — queued,
                # and booked once the NEXT step has been launched: the host stays a step ahead of the GPU (same values, same order)
                ticket = (self.flat_params.read_scalars_async(1 + len(loss_manager.losses)), validation_mode,
                          list(loss_manager.names), list(loss_manager.weights))
                for done in ([pending] if pending is not None else []):
                    values = self.flat_params.scalars(done[0])
                    loss_manager.updateLossHistory(values[1:], done[2], done[3])
                    if done[1]:
                        val_loss += values[0]
                        val_batches += 1
                    else:
                        epoch_loss += values[0]
                        epoch_batches += 1
                pending = ticket

            for done in ([pending] if pending is not None else []):  # the epoch's last step
                values = self.flat_params.scalars(done[0])
                loss_manager.updateLossHistory(values[1:], done[2], done[3])
                if done[1]:
                    val_loss += values[0]
                    val_batches += 1
                else:
                    epoch_loss += values[0]
                    epoch_batches += 1
            pending = None

            fill_stats = {}
            if fill is not None:
                # the first epoch boundary, on every rank (all ranks run the same number of steps per epoch): the rest of the own
                # slice, then the exchange — and the loaders of ALL ranks switch to indices here, or none does
                if not fill.finish(data_loader):
                    printYellow("resident frames: a rank could not complete its slice; every rank keeps decoding")
                fill_stats, fill = fill.stats, None
            elif resident is not None and epoch == 0 and not data_loader.index_mode.is_set():
                data_loader.keepPixels()

            steps = epoch_batches + val_batches
            self.epoch_stats.append(dict({"epoch": epoch + 1, "seconds": time.time() - epoch_t0, "minibatches": steps,
                                          "images": 2 * steps * self.batch_size,
                                          "index_minibatches": (resident.gathers - epoch_gathers) if resident is not None else 0},
                                         **fill_stats))
            train_loss = epoch_loss / float(max(epoch_batches, 1))
            val_loss /= float(max(val_batches, 1)) if self.world_size > 1 else float(n_val_batches)
            loss_history = loss_manager.loss_history
            loss_history['train_loss'].append(train_loss)
            loss_history['val_loss'].append(val_loss)
            for key in loss_history.keys():
                if key in ['train_loss', 'val_loss']:
                    continue
                loss_history[key][-1] /= max(epoch_batches, 1)
                if epoch + 1 < n_epochs:
                    loss_history[key].append(0)

            # (train_loss / val_loss are identical on every rank — see trainStep — so all ranks branch alike here)
            if val_loss < best_error:
                best_error = val_loss
                self.saveModel(best_model_path)

            if np.isnan(train_loss):
                printRed("NaN Loss, consider increasing NOISE_STD in the gaussian noise layer")
                sys.exit(NAN_ERROR)

            if (epoch + 1) % EPOCH_FLAG == 0:
                print("Epoch {:3}/{}, train_loss:{:.4f} val_loss:{:.4f}".format(epoch + 1, n_epochs, train_loss,
                                                                                val_loss))
                print("{:.2f}s/epoch".format((time.time() - start_time) / (epoch + 1)))

        if self.world_size > 1:
            th.distributed.barrier()
        self.model.load_state_dict(th.load(best_model_path, map_location=self.device))

        print("Predicting states for all the observations...")
        self.model.eval()
        t_pred = time.time()
        with th.no_grad():
            # The reference decodes every frame once more here (its test loader, learner.py:524-529).  With the dataset resident —
            # and a model that looks at the frames as they are (no occlusion, no negative view) — the frames are already home: the few
            # the minibatches never asked for (ragged tails) are decoded now, everything else is read from the store.
            if resident is not None and n_epochs > 0 and not self.use_dae and not self.use_triplets:
                missing = np.nonzero(~resident.have)[0]
                if len(missing):
                    chunks = [missing[i:i + 64] for i in range(0, len(missing), 64)]
                    tail_loader = makeTestLoader(chunks)
                    for chunk, frames in zip(chunks, tail_loader):
                        resident.absorb_indices(chunk, frames)
                    tail_loader.shutdown()
                pred_states = np.concatenate([self._predFn(self._toDevice(resident.store[int(mb[0]):int(mb[-1]) + 1]))
                                              for mb in test_minibatchlist if len(mb)], axis=0)
                self.predict_stats = {"from_store": True, "decoded_now": int(len(missing)), "seconds": time.time() - t_pred}
            else:
                pred_states = self.predStatesWithDataLoader(makeTestLoader())
                self.predict_stats = {"from_store": False, "decoded_now": int(len(images_path)), "seconds": time.time() - t_pred}
        pairs_loss_weight = [k for k in zip(loss_manager.names, loss_manager.weights)]
        return loss_history, pred_states, pairs_loss_weight
