"""The whole learn() loop of the product against the UNMODIFIED reference loop (GPU).

tools/make_golden.py::loop_case ran the reference's own `SRL4robotics.learn()` (models/learner.py:259-579: forked loader
process + queue, shuffled minibatches, train/validation split, `loss_history` bookkeeping, best-model checkpoint, state
prediction with the reloaded best model) on the generated dataset of tests/dataset_util.py and stored `loss_history`, the
returned states, the (name, weight) pairs and a digest of the saved checkpoint.  The product's learn() on the same
dataset, seed and hyper-parameters must reproduce them: the same minibatches in the same order (numpy RNG consumed exactly
as the reference consumes it, including in the forked loader), the same validation split, the same epoch picked as "best".

Tolerances: losses are epoch sums of up to 12 minibatches after up to 20 Adam steps -> 1e-3; learned states 2e-2 of the
state range; the checkpoint's parameters in the Adam-travel metric of golden_util.endpoint_errors (an Adam trajectory is
chaotic at rounding level, see tests/test_trajectory_gpu.py: BatchNorm biases start at zero and take +-lr steps).
"""
import json
import os

import numpy as np
import pytest
import torch

import golden_util as gu
from dataset_util import make_dataset

pytestmark = pytest.mark.gpu
HIST_RTOL = 1e-3
STATE_RTOL = 2e-2  # eval-mode states after 10-20 Adam steps are chaos-limited: the reference's own 1-thread and 8-thread
                   # runs of loop_aeif differ by 4e-3 here, its B = 2 fixtures by 4.8e-2 after 10 steps (trajectory_spread.json)
PARAM_TOL = 2e-2   # endpoint_errors metric after ~10-20 Adam steps (tests/golden/trajectory_spread.json: 1.2e-2 at 10 steps)


@pytest.mark.parametrize("name", ["loop_aeif", "loop_ae_reward"])
def test_learn_loop_follows_reference(name, tmp_path):
    import models.learner as learner
    import preprocessing.preprocess as pre
    g = gu.load(name)
    cfg = json.loads(str(g["config"]))
    ctor = cfg["ctor"]
    ds, paths, actions, rewards, starts = make_dataset(str(tmp_path), n_episodes=cfg["n_episodes"], ep_len=cfg["ep_len"])
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        os.makedirs("logs/run", exist_ok=True)
        pre.N_CHANNELS = 3
        learner.DISPLAY_PLOTS, learner.N_EPOCHS = False, cfg["n_epochs"]
        learner.BATCH_SIZE, learner.VALIDATION_SIZE = cfg["bs"], 0.2
        srl = learner.SRL4robotics(cfg["S"], model_type="custom_cnn", seed=cfg["seed"], learning_rate=cfg["lr"], cuda=True,
                                   losses=cfg["losses"], n_actions=6, log_folder="logs/run", **ctor)
        loss_history, states, pairs = srl.learn(paths, actions, rewards, starts)
        sd = torch.load("logs/run/srl_model.pth", map_location="cpu")
    finally:
        os.chdir(cwd)

    assert [p[0] for p in pairs] == [str(n) for n in g["pairs/names"]]
    np.testing.assert_allclose([float(p[1]) for p in pairs], g["pairs/weights"], rtol=0, atol=0)
    names = [str(n) for n in g["history/names"]]
    assert sorted(loss_history.keys()) == names
    worst = {}
    for nm, ref in zip(names, g["history/values"]):
        got = np.asarray(loss_history[nm], dtype=np.float64)
        assert got.shape == ref.shape, (nm, got, ref)
        err = float(np.abs(got - ref).max() / np.abs(ref).max())
        worst["history/" + nm] = err
        assert abs(got[0] - ref[0]) <= HIST_RTOL * abs(ref[0]), (nm, got, ref)   # first epoch (10 Adam steps)
        assert err <= 5 * HIST_RTOL, (nm, got, ref)                              # later epochs: chaos-limited, see above
    ref_states = g["states/full"]
    assert states.shape == ref_states.shape
    err = float(np.abs(states - ref_states).max() / np.abs(ref_states).max())
    worst["states"] = err
    assert err <= STATE_RTOL, err
    # the checkpoint on disk is the reference's best epoch (loop_aeif: validation loss rises in epoch 2 -> epoch 1 is kept)
    assert list(sd.keys()) == [str(k) for k in g["final/names"]]
    steps = cfg["n_epochs"] * 10  # optimiser steps taken (10 training minibatches per epoch)
    perr = 0.0
    for k, ref_sum, ref_abs in zip(g["final/names"], g["final/sums"], g["final/abss"]):
        k = str(k)
        v = sd[k].double()
        if "num_batches_tracked" in k:
            assert int(v) == int(ref_sum), k
        elif k not in gu.NOISE_BIASES:
            e = max(abs(float(v.sum()) - ref_sum), abs(float(v.abs().sum()) - ref_abs)) / (ref_abs + cfg["lr"] * steps * v.numel())
            perr = max(perr, e)
            assert e <= PARAM_TOL, (k, e)
    worst["params"] = perr
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/trajectory_report.jsonl", "a") as f:
            f.write(json.dumps({"case": name, "worst": worst}, sort_keys=True) + "\n")
    except OSError:
        pass
