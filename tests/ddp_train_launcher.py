"""Test-only launcher of ONE rank of a multi-rank `train.py` run (tests/test_ddp_learn_gpu.py starts one per rank, the way
torch.distributed.run would: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment).

It installs observation hooks selected by environment variables and then executes srl-zoo_amd/train.py as ``__main__`` in
this process, so what runs is the product's command line, `learn()` and checkpoint code, unmodified:

  SRLZ_TEST_DIGEST_DIR=<dir>  every rank writes <dir>/rank<r>.json after learn() returned: its loss_history, a checksum of
                              its parameters, the log folder it used, and — per saveModel() call — the LOCAL BatchNorm
                              running statistics that went into optim.average_running_stats and the averaged ones that
                              came out (so the test can check checkpoint = rank average)
  SRLZ_TEST_NAN_RANK=<r>      rank r's 2nd minibatch is poisoned with NaN (exit code 11 is expected on EVERY rank)
"""
import json
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
PKG = os.path.join(REPO, "srl-zoo_amd")
for p in (PKG, REPO, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import models.learner as learner  # noqa: E402
from srlz import optim  # noqa: E402

RANK = int(os.environ.get("RANK", "0"))
digest_dir = os.environ.get("SRLZ_TEST_DIGEST_DIR")
nan_rank = os.environ.get("SRLZ_TEST_NAN_RANK")

record = {"rank": RANK, "saves": []}

if digest_dir:
    real_avg = optim.average_running_stats

    def spy_avg(state_dict):
        key = "model.encoder_conv.1.running_mean"
        local = state_dict[key].detach().double().cpu().tolist()
        local_var = state_dict["model.encoder_conv.1.running_var"].detach().double().cpu().tolist()
        out = real_avg(state_dict)
        record["saves"].append({"local_mean": local, "local_var": local_var,
                                "avg_mean": out[key].detach().double().cpu().tolist(),
                                "tracked": int(out["model.encoder_conv.1.num_batches_tracked"])})
        return out
    optim.average_running_stats = spy_avg

    real_learn = learner.SRL4robotics.learn

    def spy_learn(self, *a, **k):
        loss_history, states, pairs = real_learn(self, *a, **k)
        flat = self.flat_params.flat.detach().double()
        record.update(loss_history={n: [float(v) for v in vals] for n, vals in loss_history.items()},
                      param_sum=float(flat.sum()), param_abs_sum=float(flat.abs().sum()),
                      log_folder=self.log_folder, world=self.world_size, adam_steps=self.optimizer.steps(),
                      states_shape=list(states.shape), states_finite=bool(np.isfinite(states).all()),
                      epoch_stats=list(getattr(self, "epoch_stats", [])),
                      loader_placement=getattr(self, "loader_placement", None),
                      resident_complete=bool(self._resident is not None and self._resident.complete()
                                             and self._resident.have.all()),
                      backend=torch.distributed.get_backend() if self.world_size > 1 else None)
        with open(os.path.join(digest_dir, "rank%d.json" % RANK), "w") as f:
            json.dump(record, f)
        return loss_history, states, pairs
    learner.SRL4robotics.learn = spy_learn

if nan_rank is not None and int(nan_rank) == RANK:
    real_step = learner.SRL4robotics.trainStep
    calls = [0]

    def poisoned(self, obs, next_obs, *a, **k):
        calls[0] += 1
        if calls[0] == 2:
            from srlz import ops
            obs, next_obs = ops.frames_as_float(obs).clone(), ops.frames_as_float(next_obs)  # (learn() feeds the loader's bytes)
            obs[0, 0, 0, 0] = float("nan")
        return real_step(self, obs, next_obs, *a, **k)
    learner.SRL4robotics.trainStep = poisoned

sys.argv = [os.path.join(PKG, "train.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
