"""Command-line trainer — same flags, loss mini-language, outputs and exit codes as the reference train.py:23-212,
driving the MI355X-native SRL4robotics.  Multi-GPU: launch one process per GPU with
``python -m torch.distributed.run --nproc-per-node N train.py ...`` (RANK / LOCAL_RANK / WORLD_SIZE are read from the
environment; backend "nccl" is RCCL on ROCm).

Out of scope of this build (rejected with a clear message): --model-type other than custom_cnn, the losses of other
SRL methods (priors, triplet, episode-prior, reward-prior), plots.
"""
from __future__ import print_function, division, absolute_import

import argparse
import json
import os
from collections import OrderedDict

import numpy as np
import torch as th

import preprocessing
import preprocessing.preprocess  # noqa: F401  (N_CHANNELS is set below)
import models.learner as learner
from models.learner import SRL4robotics
from pipeline import getLogFolderName, saveConfig, correlationCall
from srlz import optim
from utils import parseDataFolder, createFolder, loadData, buildConfig, parseLossArguments

LOSS_CHOICES = ["forward", "inverse", "reward", "priors", "episode-prior", "reward-prior", "triplet",
                "autoencoder", "vae", "perceptual", "dae", "random"]


# The reference's command line (train.py:25-61): identical flags, short forms, types and defaults; one row per flag.
_INT, _FLOAT, _STR, _FLAG = int, float, str, "flag"
_CLI = [
    (("--epochs",), _INT, 30, "training epochs"),
    (("--seed",), _INT, 1, "seed of numpy / torch"),
    (("--state-dim",), _INT, 2, "dimension of the learned state"),
    (("-bs", "--batch-size"), _INT, 32, "samples per minibatch (per GPU)"),
    (("--val-size",), _FLOAT, 0.2, "fraction of the minibatches held out for validation"),
    (("--training-set-size",), _INT, -1, "use only the first N samples (-1: all)"),
    (("-lr", "--learning-rate"), _FLOAT, 0.005, "Adam learning rate"),
    (("--l1-reg",), _FLOAT, 0.0, "weight of the L1 regulariser"),
    (("--l2-reg",), _FLOAT, 0.0, "weight of the L2 regulariser"),
    (("--no-cuda",), _FLAG, False, "accepted for compatibility; this build has no CPU path"),
    (("--no-display-plots",), _FLAG, False, "accepted for compatibility; plotting is not part of this build"),
    (("--data-folder",), _STR, "", "dataset folder under data/ (required)"),
    (("--log-folder",), _STR, "", "output folder (default: logs/<dataset>/<timestamp>_<model>_ST_DIM<n>_<losses>)"),
    (("--multi-view",), _FLAG, False, "two stacked camera views (6 input channels)"),
    (("--balanced-sampling",), _FLAG, False, "accepted for compatibility (episode prior only)"),
    (("--beta",), _FLOAT, 1.0, "weight of the KL term (beta-VAE)"),
    (("--path-to-dae",), _STR, "", "srl_model.pth of a trained DAE (perceptual loss)"),
    (("--state-dim-dae",), _INT, 200, "state dimension of that DAE"),
    (("--occlusion-percentage",), _FLOAT, 0.5, "largest occluded fraction per side (DAE)"),
]


def buildParser():
    parser = argparse.ArgumentParser(description='State Representation Learning on MI355X (srl-zoo hot path)')
    for names, kind, default, text in _CLI:
        if kind == _FLAG:
            parser.add_argument(*names, action='store_true', default=default, help=text)
        else:
            parser.add_argument(*names, type=kind, default=default, help="%s (default: %r)" % (text, default),
                                required=(names[0] == "--data-folder"))
    parser.add_argument('--model-type', type=str, default="custom_cnn", choices=['custom_cnn', 'resnet', 'mlp', 'linear'],
                        help='encoder family; only custom_cnn runs here')
    parser.add_argument('--inverse-model-type', type=str, default="linear", choices=['mlp', 'linear'],
                        help='architecture of the inverse model')
    parser.add_argument('--losses', nargs='+', default=["inverse"], **parseLossArguments(
        choices=LOSS_CHOICES, help='losses to combine, as <name> or <name>:<weight>[:<dimension>]'))
    return parser


def resolveLosses(raw_losses, multi_view):
    """The loss mini-language of the reference (train.py:79-115): plain names, or all '<name>:<weight>[:<dim>]'.
    :return: (losses, losses_weights_dict, split_dimensions)"""
    described = [isinstance(loss, tuple) for loss in raw_losses]
    if any(described) and not all(described):
        raise ValueError("Either no losses have a defined weight or dimension, or all losses have a defined weight. "
                         "{}".format(raw_losses))
    if not all(described):
        return list(set(raw_losses)), None, -1
    losses_weights_dict, split_dimensions = OrderedDict(), OrderedDict()
    for loss, weight, split_dim in raw_losses:
        losses_weights_dict[loss] = weight
        split_dimensions[loss] = split_dim
    losses = list(losses_weights_dict.keys())
    assert not ("triplet" in losses and not multi_view), \
        "Triplet loss with single view is not supported, please use the --multi-view option"
    return losses, losses_weights_dict, split_dimensions


def initDistributed():
    """One process per GPU when launched by torch.distributed.run; returns (rank, world_size)."""
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size == 1:
        return 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    th.cuda.set_device(optim.local_device_index())
    th.distributed.init_process_group(backend=optim.dist_backend())  # "nccl" = RCCL over xGMI
    placement = optim.rank_devices()  # (collective; under RCCL: one distinct GPU per rank, or every rank stops here with a message)
    if th.distributed.get_rank() == 0:
        print("{} ranks on {} GPU(s) of {} host(s), backend {}".format(placement["ranks"], placement["devices"], placement["hosts"],
                                                                     th.distributed.get_backend()))
    if optim.native_comm_requested():  # SRLZ_COMM=rccl: the bucket travels through srlz_comm_allreduce_f32 (include/srlz.h)
        optim.init_native_comm()
    return th.distributed.get_rank(), world_size


if __name__ == '__main__':
    args = buildParser().parse_args()
    args.cuda = not args.no_cuda and th.cuda.is_available()
    args.data_folder = parseDataFolder(args.data_folder)
    learner.DISPLAY_PLOTS = False  # plotting is out of scope
    learner.N_EPOCHS = args.epochs
    learner.BATCH_SIZE = args.batch_size
    learner.VALIDATION_SIZE = args.val_size
    learner.BALANCED_SAMPLING = args.balanced_sampling

    losses, losses_weights_dict, split_dimensions = resolveLosses(args.losses, args.multi_view)
    args.losses = losses
    args.split_dimensions = split_dimensions
    if args.multi_view is True:
        # two stacked camera views (three with triplets) -> input layers take 6 (9) channels
        preprocessing.preprocess.N_CHANNELS = 9 if "triplet" in losses else 6

    assert not ("autoencoder" in losses and "vae" in losses), "Model cannot be both an Autoencoder and a VAE (come on!)"
    assert not (("autoencoder" in losses or "vae" in losses)
                and args.model_type == "resnet"), "Model cannot be an Autoencoder or VAE using ResNet Architecture !"
    assert not ("vae" in losses and args.model_type == "linear"), "Model cannot be VAE using Linear Architecture !"
    assert not (args.multi_view and args.model_type == "resnet"), \
        "Default ResNet input layer is not suitable for stacked images!"
    assert not (args.path_to_dae == "" and "vae" in losses and "perceptual" in losses), \
        "To use the perceptual loss with a VAE, please specify a path to a pre-trained DAE model"
    assert not ("dae" in losses and "perceptual" in losses), \
        "Please learn the DAE before learning a VAE with the perceptual loss "

    rank, world_size = initDistributed()

    print('Loading data ... ')
    training_data, ground_truth, _, _ = loadData(args.data_folder)
    rewards, episode_starts = training_data['rewards'], training_data['episode_starts']
    actions = training_data['actions']
    n_actions = int(np.max(actions) + 1)  # actions are assumed to be integers
    try:
        images_path = np.array([path.decode("utf-8") for path in ground_truth['images_path']])
    except AttributeError:
        images_path = ground_truth['images_path']

    exp_config = buildConfig(args)
    # the folder name carries a wall-clock timestamp: rank 0 decides and creates, the other ranks are told
    if rank == 0:
        if args.log_folder == "":
            createFolder("logs/{}".format(exp_config['data-folder']), "Dataset log folder already exist")
            log_folder, experiment_name = getLogFolderName(exp_config)
        else:
            log_folder = args.log_folder
            createFolder(log_folder, "Log folder already exist")
            experiment_name = "{}_{}".format(args.model_type, losses)
    else:
        log_folder = experiment_name = None
    args.log_folder, experiment_name = optim.share_from_rank0((log_folder, experiment_name))

    exp_config['log-folder'] = args.log_folder
    exp_config['experiment-name'] = experiment_name
    exp_config['n_actions'] = n_actions
    exp_config['multi-view'] = args.multi_view
    if "dae" in losses:
        exp_config['occlusion-percentage'] = args.occlusion_percentage
    print('Log folder: {}'.format(args.log_folder))

    print('Learning a state representation ... ')
    srl = SRL4robotics(args.state_dim, model_type=args.model_type, inverse_model_type=args.inverse_model_type,
                       seed=args.seed, log_folder=args.log_folder, learning_rate=args.learning_rate,
                       l1_reg=args.l1_reg, l2_reg=args.l2_reg, cuda=args.cuda, multi_view=args.multi_view,
                       losses=losses, losses_weights_dict=losses_weights_dict, n_actions=n_actions, beta=args.beta,
                       split_dimensions=split_dimensions, path_to_dae=args.path_to_dae,
                       state_dim_dae=args.state_dim_dae, occlusion_percentage=args.occlusion_percentage)

    if args.training_set_size > 0:
        limit = args.training_set_size
        actions, images_path = actions[:limit], images_path[:limit]
        rewards, episode_starts = rewards[:limit], episode_starts[:limit]

    if rank == 0:
        saveConfig(exp_config, print_config=True)

    loss_history, learned_states, pairs_name_weights = srl.learn(images_path, actions, rewards, episode_starts)

    if rank == 0:
        exp_config['losses_weights'] = pairs_name_weights
        saveConfig(exp_config, print_config=True)
        srl.saveStates(learned_states, images_path, rewards, args.log_folder)
        np.savez('{}/loss_history.npz'.format(args.log_folder), **loss_history)
        with open('{}/epoch_stats.json'.format(args.log_folder), 'w') as f:  # (build-specific: per-epoch wall time and frames)
            json.dump(getattr(srl, "epoch_stats", []), f)
        with open('{}/predict_stats.json'.format(args.log_folder), 'w') as f:  # (where the final states' frames came from, seconds)
            json.dump(getattr(srl, "predict_stats", {}), f)
        correlationCall(exp_config, plot=False)
    if world_size > 1:
        # (every rank's own per-epoch record: wall seconds, minibatches served from the resident store, the slice exchange)
        with open('{}/epoch_stats_rank{}.json'.format(args.log_folder, rank), 'w') as f:
            json.dump(getattr(srl, "epoch_stats", []), f)
        optim.destroy_native_comm()  # (no-op unless SRLZ_COMM=rccl created the library's own communicator)
        th.distributed.destroy_process_group()
