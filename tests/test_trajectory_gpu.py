"""The optimisation TRAJECTORY of the product against the unmodified reference (GPU).

`SRL4robotics.trainStep` — the product's own loop body, not a test-local restatement — is driven over the inputs of the
committed multi-step fixtures (tools/make_golden.py: the reference's loop body models/learner.py:360-498 with
th.optim.Adam, lr 1e-4, inputs seed 1234+step, VAE noise from th.manual_seed(99+step)):

    zero_grad -> forward x2 (x4 for the VAE quirk) -> losses -> backward -> gradient delivery (fold) -> Adam
      -> the NEXT step's BatchNorm state / parameters,

including validation minibatches in the middle of a trajectory (eval mode, backward run and discarded, no Adam step:
reference learner.py:362-364,487-497) and the l1+l2 case that overflows the gradient staging buckets (4 contributions per
encoder weight, srlz/optim.py NSTAGE=3).

Checked per case: every per-step loss (5e-4, the tolerance the CPU twin is held to), every parameter's sum / abs-sum after
the last step, the BatchNorm running statistics and counters, and the eval-mode states of the updated model.

Pre-BatchNorm ConvTranspose biases (decoder_conv.{0,3,6,9}.bias) have an analytically ZERO gradient; what either
implementation feeds Adam there is summation noise, which Adam's normalisation turns into +-lr steps of random sign.  They
cannot agree and do not matter (BatchNorm removes a per-channel constant): bounded by lr * steps per element instead.
"""
import json
import os

import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu
LR = 1e-4
LOSS_RTOL = 5e-4
PARAM_RTOL = 2e-4   # |sum - ref| and |abs-sum - ref| relative to the reference abs-sum
BN_RTOL = 1e-4
STATE_RTOL = 5e-4
NOISE_BIASES = tuple("model.decoder_conv.%d.bias" % i for i in (0, 3, 6, 9))

_split = gu.ext_defaults(gu.ext_cases()["step_split_dae_rfi_b4"])
CASES = {
    "trace_ae_b2": dict(losses=["autoencoder"], n_steps=3),
    "trace_vae_b2": dict(losses=["vae"], n_steps=3),
    "trace_aeif_b2": dict(losses=["autoencoder", "inverse", "forward"], n_steps=3),
    "trace_split_dae_rfi_b4": dict(losses=_split["losses"], n_steps=3, B=4, S=_split["S"], inverse=_split["inverse"],
                                   split=_split["split"], weights=_split["weights"], l2_reg=_split["l2_reg"]),
    "trace10_ae_b2": dict(losses=["autoencoder"], n_steps=10),
    "trace10_vae_b2": dict(losses=["vae"], n_steps=10),
    "trace_val_aeif_b2": dict(losses=["autoencoder", "inverse", "forward"], n_steps=4, val_steps=(1,)),
    "trace_val_vae_b2": dict(losses=["vae"], n_steps=4, val_steps=(2,)),
    "trace_ae_l1l2_b2": dict(losses=["autoencoder"], n_steps=3, l1_reg=1e-5, l2_reg=1e-4),
}


def drive_product(losses, n_steps, B=2, S=200, inverse="linear", split=None, weights=None, l1_reg=0.0, l2_reg=0.0,
                  val_steps=()):
    """n_steps calls of SRL4robotics.trainStep on the fixture inputs; returns (learner, [per-step {name: value}])."""
    import models.learner as learner
    import preprocessing.preprocess as pre
    from losses.losses import LossManager
    pre.N_CHANNELS = 3
    learner.BATCH_SIZE = B
    srl = learner.SRL4robotics(S, model_type="custom_cnn", inverse_model_type=inverse, seed=1, learning_rate=LR, cuda=True,
                               losses=losses, losses_weights_dict=weights, n_actions=6, log_folder="/tmp",
                               split_dimensions=split if split is not None else -1, l1_reg=l1_reg, l2_reg=l2_reg)
    dev = srl.device
    if "vae" in losses:  # the reference draws eps from the CPU generator (models/models.py:161); same draws, same order
        srl.model.model.eps_fn = lambda mu: torch.empty(mu.shape).normal_().to(mu.device)
    lm = LossManager(srl.model, None)
    trace = []
    for step in range(n_steps):
        obs, nxt, act = gu.golden_inputs(B, 3, 6, seed=1234 + step)
        noisy = next_noisy = rew = None
        if "dae" in losses:
            noisy = torch.from_numpy(gu.golden_noisy(obs, seed=1234 + step)).to(dev)
            next_noisy = torch.from_numpy(gu.golden_noisy(nxt, seed=4321 + step)).to(dev)
        if "reward" in losses:
            rew = torch.from_numpy(gu.golden_rewards(B, seed=1234 + step)[1]).to(dev)
        torch.manual_seed(99 + step)
        loss = srl.trainStep(torch.from_numpy(obs).to(dev), torch.from_numpy(nxt).to(dev),
                             torch.from_numpy(act).view(-1, 1).to(dev), lm, validation_mode=step in val_steps,
                             noisy_obs=noisy, next_noisy_obs=next_noisy, rewards_st=rew)
        rec = dict(zip(lm.names, lm.lossValues()))
        rec["total"] = float(loss.detach())
        trace.append(rec)
    torch.cuda.synchronize()
    return srl, trace


def compare_with_fixture(name, srl, trace, n_steps):
    """-> (list of failure strings, dict of worst errors)."""
    g = gu.load(name)
    fails, worst = [], {}

    def note(kind, err, tol, what):
        worst[kind] = max(worst.get(kind, 0.0), err)
        if not err <= tol:
            fails.append("%s: %s err %.3e > %.1e" % (kind, what, err, tol))

    names = [str(n) for n in g["trace/names"]]
    assert g["trace/values"].shape == (n_steps, len(names))
    for step, rec in enumerate(trace):
        assert sorted(rec.keys()) == sorted(names), (sorted(rec.keys()), names)
        for j, nm in enumerate(names):
            v = float(g["trace/values"][step, j])
            note("loss", abs(rec[nm] - v) / max(abs(v), 1e-6), LOSS_RTOL, "step %d %s (%.6g vs %.6g)" % (step, nm, rec[nm], v))

    sd = srl.model.state_dict()
    assert [str(k) for k in g["final/names"]] == list(sd.keys())
    for k, ref_sum, ref_abs in zip(g["final/names"], g["final/sums"], g["final/abss"]):
        k = str(k)
        v = sd[k].detach().double().cpu()
        if "num_batches_tracked" in k:
            assert int(v) == int(ref_sum), (k, int(v), int(ref_sum))
            continue
        e_sum = abs(float(v.sum()) - ref_sum)
        e_abs = abs(float(v.abs().sum()) - ref_abs)
        if k in NOISE_BIASES:  # +-lr per step and element at most (see the module docstring)
            bound = 2.0 * LR * n_steps * v.numel()
            note("noise_bias", max(e_sum, e_abs) / bound, 1.0, k)
        else:
            note("param", max(e_sum, e_abs) / max(ref_abs, 1e-30), PARAM_RTOL, k)
    for k in [f for f in g.files if f.startswith("final_bn/")]:
        key = k[len("final_bn/"):]
        ref = g[k]
        v = sd[key].detach().double().cpu().numpy()
        if "num_batches" in key:
            assert int(v) == int(ref), (key, int(v), int(ref))
        else:
            note("bn", float(np.abs(v - ref).max()) / max(float(np.abs(ref).max()), 1e-30), BN_RTOL, key)

    srl.model.eval()
    obs, _, _ = gu.golden_inputs(int(g["eval_states/full"].shape[0]), 3, 6, seed=1234)
    with torch.no_grad():
        st = srl.model.getStates(torch.from_numpy(obs).to(srl.device)).double().cpu().numpy()
    ref = g["eval_states/full"]
    note("eval_states", float(np.abs(st - ref).max()) / float(np.abs(ref).max()), STATE_RTOL, "eval-mode states")
    return fails, worst


def _report(name, worst):
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open(os.path.join("gpurun_out", "trajectory_report.jsonl"), "a") as f:
            f.write(json.dumps({"case": name, "worst": worst}, sort_keys=True) + "\n")
    except OSError:
        pass


@pytest.mark.parametrize("name", sorted(CASES))
def test_train_step_trajectory_follows_reference(name):
    cfg = dict(CASES[name])
    srl, trace = drive_product(**cfg)
    # an optimiser step was taken for every training minibatch and none for a validation one
    assert srl.optimizer.steps() == cfg["n_steps"] - len(cfg.get("val_steps", ()))
    fails, worst = compare_with_fixture(name, srl, trace, cfg["n_steps"])
    _report(name, worst)
    assert not fails, "\n".join(fails)


def test_staging_overflow_is_exercised():
    """l1 + l2 + two frames = four gradient contributions per regularised weight: the fourth finds no staging bucket
    (FlatParams.grad_buffer -> None) and must travel through autograd's own accumulation."""
    import models.learner as learner
    from srlz import optim
    calls = {"none": 0}
    orig = optim.FlatParams.grad_buffer

    def spy(self, index):
        buf = orig(self, index)
        if buf is None:
            calls["none"] += 1
        return buf
    optim.FlatParams.grad_buffer = spy
    try:
        drive_product(["autoencoder"], 1, l1_reg=1e-5, l2_reg=1e-4)
    finally:
        optim.FlatParams.grad_buffer = orig
    assert calls["none"] > 0
