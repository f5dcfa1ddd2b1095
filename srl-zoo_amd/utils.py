"""CLI / config helpers of the trainer (reference utils.py:14-210): loss-spec argparse type, experiment config,
dataset loading, tensor -> numpy, folder and coloured-print helpers."""
from __future__ import print_function, division

import argparse
import json
import os
from collections import OrderedDict

import numpy as np
import torch as th

_COLORS = {"red": "\033[31m", "green": "\033[32m", "yellow": "\033[33m", "blue": "\033[34m"}


def parseLossArguments(choices, help):
    """argparse `type` for ``--losses``: accepts ``<name>``, ``<name>:<weight>`` or ``<name>:<weight>:<dimension>``.

    'autoencoder:1:10' -> ('autoencoder', 1.0, 10);  'inverse' -> 'inverse'.
    :return: (dict) kwargs for parser.add_argument
    """
    def _arg_type(arg):
        fields = arg.split(':')
        if fields[0] not in choices:
            raise argparse.ArgumentTypeError("invalid choice: {} (choose from {})".format(fields[0], choices))
        if len(fields) == 1:
            return arg
        try:
            weight = float(fields[1])
            dimension = int(fields[2]) if len(fields) == 3 else 0
        except (ValueError, IndexError):
            raise argparse.ArgumentTypeError(
                "Error: must be of format '<str>:<float>:<int>', '<str>:<float/int>' or '<str>'")
        if len(fields) > 3:
            raise argparse.ArgumentTypeError(
                "Error: must be of format '<str>:<float>:<int>', '<str>:<float/int>' or '<str>'")
        return fields[0], weight, dimension

    return {'type': _arg_type, 'help': "{" + ", ".join(choices) + "} " + help}


def buildConfig(args):
    """Experiment config (exp_config.json) from parsed arguments; key set and order follow the reference."""
    get = lambda name, default: getattr(args, name) if hasattr(args, name) else default
    if "supervised" in args.losses:
        args.inverse_model_type = None
    return OrderedDict([
        ("batch-size", args.batch_size),
        ("beta", get("beta", -1)),
        ("data-folder", args.data_folder),
        ("epochs", args.epochs),
        ("learning-rate", args.learning_rate),
        ("training-set-size", args.training_set_size),
        ("log-folder", ""),
        ("model-type", args.model_type),
        ("seed", args.seed),
        ("state-dim", args.state_dim),
        ("knn-samples", 200),
        ("knn-seed", 1),
        ("l1-reg", get("l1_reg", 0)),
        ("l2-reg", get("l2_reg", 0)),
        ("losses", args.losses),
        ("n-neighbors", 5),
        ("n-to-plot", 5),
        ("split-dimensions", get("split_dimensions", -1)),
        ("inverse-model-type", args.inverse_model_type),
    ])


def loadData(data_folder):
    """Load data/<folder>/{preprocessed_data.npz, ground_truth.npz, dataset_config.json}.

    :return: (training_data, ground_truth, true_states, target_positions) — true states are made relative to the
             episode's target when the dataset config says so.
    """
    training_data = np.load('data/{}/preprocessed_data.npz'.format(data_folder))
    episode_starts = training_data['episode_starts']
    ground_truth = np.load('data/{}/ground_truth.npz'.format(data_folder))
    keys = list(ground_truth.keys())
    true_states = ground_truth['ground_truth_states' if 'ground_truth_states' in keys else 'arm_states']
    target_positions = ground_truth['target_positions' if 'target_positions' in keys else 'button_positions']
    with open('data/{}/dataset_config.json'.format(data_folder), 'r') as f:
        relative_pos = json.load(f).get('relative_pos', False)

    episode_idx = np.cumsum(np.asarray(episode_starts) == 1) - 1
    per_frame_target = np.array([target_positions[i] for i in episode_idx])
    if relative_pos:
        true_states = true_states - per_frame_target
    return training_data, ground_truth, true_states, per_frame_target


def getInputBuiltin():
    try:
        return raw_input  # noqa: F821 (python 2)
    except NameError:
        return input


def detachToNumpy(tensor):
    """th.Tensor (any device) -> np.ndarray."""
    return tensor.to(th.device('cpu')).detach().numpy()


def parseDataFolder(path):
    """Strip a leading 'data/' and a trailing '/' from a dataset path."""
    if path.startswith('data/'):
        path = path[len('data/'):]
    return path.rstrip('/')


def createFolder(path, exist_msg):
    try:
        os.makedirs(path)
    except OSError:
        print(exist_msg)


def _cprint(color, string):
    print(_COLORS[color] + str(string) + "\033[0m")


def printGreen(string):
    _cprint("green", string)


def printYellow(string):
    _cprint("yellow", string)


def printRed(string):
    _cprint("red", string)


def printBlue(string):
    _cprint("blue", string)
