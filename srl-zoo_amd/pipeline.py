"""The three helpers train.py imports from the reference's pipeline.py (getLogFolderName :28-51, saveConfig :223-236,
correlationCall) plus the exit-code constants (:23-25).  The grid-search driver itself is out of scope."""
from __future__ import print_function, division

import datetime
import json
from collections import OrderedDict
from pprint import pprint

from utils import printBlue, printYellow, createFolder

MATPLOTLIB_WARNING_CODE = -11
NO_PAIRS_ERROR = 10  # no dissimilar/reference pairs found (robotic priors)
NAN_ERROR = 11       # training loss became NaN


def getLogFolderName(exp_config):
    """logs/<dataset>/<YY-MM-DD_HHhMM_SS>_<model>_ST_DIM<S>_<losses>; creates the folder.
    :return: (log_folder, experiment_name)"""
    date = datetime.datetime.now().strftime("%y-%m-%d_%Hh%M_%S")
    losses = exp_config["losses"]
    if not isinstance(losses, str):
        losses = "_".join(losses)
    experiment_name = "{}_{}_ST_DIM{}_{}".format(date, exp_config['model-type'], exp_config['state-dim'], losses)
    printBlue("\nExperiment: {}\n".format(experiment_name))
    log_folder = "logs/{}/{}".format(exp_config['data-folder'], experiment_name)
    createFolder(log_folder, "Experiment folder already exist")
    return log_folder, experiment_name


def saveConfig(exp_config, print_config=False):
    """Write <log-folder>/exp_config.json (keys sorted)."""
    if print_config:
        pprint(exp_config)
    exp_config = OrderedDict(sorted(exp_config.items()))
    with open("{}/exp_config.json".format(exp_config['log-folder']), "w") as f:
        json.dump(exp_config, f)
    print("Saved config to log folder: {}".format(exp_config['log-folder']))


def correlationCall(exp_config, plot=False):
    """The reference shells out to plotting.representation_plot for the ground-truth correlation; plotting and
    evaluation are outside this build's scope, so this only says so."""
    printYellow("correlationCall: ground-truth correlation (plotting/) is out of scope of the MI355X hot-path build")
