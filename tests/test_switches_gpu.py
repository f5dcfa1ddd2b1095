"""Every environment switch that selects a non-default code path takes two training steps in a fresh process and must
land where the default path lands: same losses, same parameters after Adam (to summation order — the switched paths run other
kernels or other launch structures of the same arithmetic).  A switch nothing selects is a path the next kernel change breaks
silently."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import json, os, sys
sys.path[:0] = [os.path.join(sys.argv[1], "srl-zoo_amd"), sys.argv[1], os.path.join(sys.argv[1], "tests")]
import numpy as np, torch
import golden_util as gu
import models.learner as learner
import preprocessing.preprocess as pre
from losses.losses import LossManager
pre.N_CHANNELS = 3
learner.BATCH_SIZE = 3
losses = sys.argv[2].split(",")
srl = learner.SRL4robotics(24, model_type="custom_cnn", seed=7, learning_rate=1e-3, cuda=True, losses=losses, n_actions=6, log_folder="/tmp")
lm = LossManager(srl.model, None)
rows = []
for step in range(2):
    o, n, a = gu.golden_inputs(3, 3, 6, seed=500 + step)
    both = torch.from_numpy(np.concatenate((o, n), 0)).cuda()
    if "vae" in losses:
        torch.manual_seed(90 + step)
        it = iter([torch.randn(3, 24), torch.randn(3, 24)])
        srl.model.model.eps_fn = lambda mu: next(it).to(mu.device)
    loss = srl.trainStep(both[:3], both[3:], torch.from_numpy(a).view(-1, 1).cuda(), lm)
    rows.append(lm.lossValues() + [float(loss.detach())])
torch.cuda.synchronize()
flat = srl.flat_params.flat.double().cpu().numpy()
bufs = [float(b.double().abs().sum()) for b in srl.model.buffers()]  # (no cancellation: a sum of running means is ~0)
np.save(sys.argv[3], flat)
print("RESULT " + json.dumps({"losses": rows, "names": list(lm.names), "bufs": bufs, "steps": srl.optimizer.steps()}))
"""

# Round 5 retired the A/B residue (twelve variables): what is left selects a feature (hipGraph replay) and a debugging
# aid (synchronise after every call).  The fallback ROUTES the retired variables used to select are reached the way the product
# reaches them — by shape or by what the step needs — in tests/test_pair_gpu.py, test_step_gpu.py, test_trajectory_gpu.py,
# test_kernels_gpu.py.
SWITCHES = [("SRLZ_GRAPH", "1"), ("SRLZ_SYNC", "1")]
RETIRED = ["SRLZ_DEFER_BN_BWD", "SRLZ_FUSE_ENC_IN", "SRLZ_DIRECT_GRADS", "SRLZ_WGRAD_RING", "SRLZ_WGRAD_S2_TK32", "SRLZ_PAIR",
           "SRLZ_FUSED_TOTAL", "SRLZ_FUSED_OUT_BWD", "SRLZ_FUSED_RECON", "SRLZ_POOL_BWD_IN_DGRAD", "SRLZ_FUSED_CONVT_BWD",
           "SRLZ_DGRAD_PIPE"]


def _run_all(losses, tmp_path):
    """Default + every switch, four processes at a time."""
    # (graph replay draws the VAE's noise inside the captured graph; this script injects the noise of each step from outside, which
    # the capture's warm-up executions would use up — the hipGraph path is compared on the deterministic configuration)
    switches = [kv for kv in SWITCHES if not (kv[0] == "SRLZ_GRAPH" and "vae" in losses)]
    jobs = [("default", {})] + [("%s=%s" % kv, dict([kv])) for kv in switches]
    results, running = {}, []

    def reap(block):
        for item in list(running):
            name, proc, path = item
            if block:
                proc.wait()
            if proc.poll() is None:
                continue
            running.remove(item)
            out = proc.stdout.read().decode("utf-8", "replace")
            assert proc.returncode == 0, "%s: exit %d\n%s" % (name, proc.returncode, out[-3000:])
            line = [l for l in out.splitlines() if l.startswith("RESULT ")][-1]
            results[name] = (json.loads(line[7:]), np.load(path))
    for name, extra in jobs:
        while len(running) >= 4:
            reap(False)
            if len(running) >= 4:
                running[0][1].wait()
        env = dict(os.environ)
        for k, _ in SWITCHES:
            env.pop(k, None)
        env.update(extra)
        path = str(tmp_path / (name.replace("=", "_") + ".npy"))
        proc = subprocess.Popen([sys.executable, "-c", _SCRIPT, REPO, ",".join(losses), path], env=env, stdout=subprocess.PIPE,
                                stderr=subprocess.STDOUT)
        running.append((name, proc, path))
    while running:
        reap(True)
    return results


@pytest.mark.timeout(900)
@pytest.mark.parametrize("losses", [["autoencoder", "inverse"], ["vae"]], ids=["ae_inverse", "vae"])
def test_every_switch_lands_where_the_default_path_lands(losses, tmp_path):
    res = _run_all(losses, tmp_path)
    ref, pref = res["default"]
    assert ref["steps"] == 2 and np.isfinite(ref["losses"]).all()
    scale = float(np.abs(pref).max())
    for name, (got, p) in sorted(res.items()):
        if name == "default":
            continue
        assert got["names"] == ref["names"] and got["steps"] == 2, name
        np.testing.assert_allclose(got["losses"], ref["losses"], rtol=3e-5, err_msg=name)
        # Adam turns a 1e-6 gradient difference at a rounding-noise element into a +-lr step: bound the parameters by a few lr
        assert float(np.abs(p - pref).max()) <= 4e-3 * scale, (name, float(np.abs(p - pref).max()), scale)
        assert float(np.abs(p - pref).mean()) <= 2e-5 * scale, (name, float(np.abs(p - pref).mean()))
        # (sums of |running_mean| / running_var over a layer's 64 channels, and the step counts.  The bias of a ConvTranspose in front of
        # a training-mode BatchNorm has a gradient that is rounding noise — the normalisation removes the mean — so Adam moves it by
        # +-lr per step on whichever side the noise falls, and the layer's batch means move with it: 64 channels x lr 1e-3 x momentum
        # 0.1 x 2 steps bounds what two summation orders can differ by in such a sum, a few 1e-3 absolute)
        np.testing.assert_allclose(got["bufs"], ref["bufs"], rtol=1e-3, atol=5e-3, err_msg=name)


def test_retired_switches_are_gone_from_the_sources():
    """No source file reads a retired variable any more (a variable that is read but untested is a path the next change breaks)."""
    import glob
    files = [f for pat in ("srl-zoo_amd/**/*.py", "srl-zoo_amd/csrc/*", "bench.py") for f in glob.glob(os.path.join(REPO, pat), recursive=True)
             if os.path.isfile(f) and not f.endswith((".o", ".so", ".pyc"))]
    assert len(files) > 20
    for f in files:
        text = open(f, errors="replace").read()
        for name in RETIRED:
            assert ('"%s"' % name) not in text, (f, name)
