"""The optimisation TRAJECTORY of the product against the unmodified reference (GPU).

`SRL4robotics.trainStep` — the product's own loop body, not a test-local restatement — is driven over the inputs of the
committed multi-step fixtures (tools/make_golden.py: the reference's loop body models/learner.py:360-498 with
th.optim.Adam, lr 1e-4, inputs seed 1234+step, VAE noise from th.manual_seed(99+step)):

    zero_grad -> forward x2 (x4 for the VAE quirk) -> losses -> backward -> gradient delivery (fold) -> Adam
      -> the NEXT step's BatchNorm state / parameters,

including validation minibatches in the middle of a trajectory (eval mode, backward run and discarded, no Adam step:
reference learner.py:362-364,487-497) and the l1+l2 case that overflows the gradient staging buckets (4 contributions per
encoder weight, srlz/optim.py NSTAGE=3).

Two tests per fixture, because an Adam trajectory at B = 2 is CHAOTIC in fp32: Adam divides every gradient element by its
own magnitude, so an element whose gradient is rounding noise (or sits next to a ReLU / max-pool tie) takes +-lr steps of
arbitrary sign.  The reference run with 8 instead of 1 MKLDNN threads ends 0.6 % (3 steps) / 4 % (10 steps) away from
itself in eval-mode states (tools/measure_spread.py -> tests/golden/trajectory_spread.json).

 1. test_train_step_trajectory_follows_reference — free running against the fixture: every per-step loss to 5e-4 (the
    tolerance the CPU twin is held to), optimiser-step and num_batches_tracked counters exactly, and the END POINT
    (parameter sums, BatchNorm running statistics, eval-mode states) within SPREAD_FACTOR x the reference's own rounding
    spread, taken as the largest one over the fixtures with as many steps (whether a particular fixture's handful of
    noisy elements flips in a particular run is luck; the amplitude when they do is the property) — a gross-divergence
    bound, no more.
 2. test_train_step_rides_along_the_oracle — the tight one, by induction over the steps: before every step the CPU oracle
    is re-seeded with the product's CURRENT parameters / buffers, so nothing accumulates.  Per step: every loss 1e-5;
    BatchNorm buffers after the step 1e-5; the gradient bucket Adam consumed against the oracle's gradient (norm 5e-3,
    direction 3e-2 — set by the handful of near-tie ReLU / max-pool decisions per step, cf. tests/test_step_gpu.py;
    measured on MI355X: losses 8e-7, buffers 5e-7, norms 7e-4, directions 1e-2, update 6e-8); the parameter update
    against Adam's formula evaluated in fp64 on that very bucket (3e-7 absolute); a validation step leaves parameters,
    moments, step count and running statistics bit-identical.

Pre-BatchNorm ConvTranspose biases (decoder_conv.{0,3,6,9}.bias) have an analytically ZERO gradient; what either
implementation feeds Adam there is summation noise.  They are reported separately and bounded by Adam's travel only.
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
SPREAD_FACTOR = 5.0
ENDPOINT_FLOOR = {"param": 2e-4, "bn": 2e-4, "eval_states": 5e-4, "noise_bias": 1.0}
NOISE_BIASES = gu.NOISE_BIASES

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
    # the reference's DEFAULT minibatch (bs = 32, BASELINE.json configs[0]): 16x the samples per gradient -> far fewer
    # rounding-noise gradient elements -> its end point is held to ITS OWN (much smaller) measured spread
    "trace10_ae_b32": dict(losses=["autoencoder"], n_steps=10, B=32),
}


def make_learner(losses, B=2, S=200, inverse="linear", split=None, weights=None, l1_reg=0.0, l2_reg=0.0, **_):
    import models.learner as learner
    import preprocessing.preprocess as pre
    pre.N_CHANNELS = 3
    learner.BATCH_SIZE = B
    return learner.SRL4robotics(S, model_type="custom_cnn", inverse_model_type=inverse, seed=1, learning_rate=LR, cuda=True,
                                losses=losses, losses_weights_dict=weights, n_actions=6, log_folder="/tmp",
                                split_dimensions=split if split is not None else -1, l1_reg=l1_reg, l2_reg=l2_reg)


def step_inputs(losses, step, B, S):
    """The fixture's inputs of one step as CPU tensors (tools/make_golden.py::step_case)."""
    obs, nxt, act = gu.golden_inputs(B, 3, 6, seed=1234 + step)
    d = dict(obs=torch.from_numpy(obs), next_obs=torch.from_numpy(nxt), actions=torch.from_numpy(act), noisy=(None, None),
             rewards=None, eps=(None, None))
    if "dae" in losses:
        d["noisy"] = (torch.from_numpy(gu.golden_noisy(obs, seed=1234 + step)),
                      torch.from_numpy(gu.golden_noisy(nxt, seed=4321 + step)))
    if "reward" in losses:
        d["rewards"] = torch.from_numpy(gu.golden_rewards(B, seed=1234 + step)[1])
    if "vae" in losses:  # the reference draws eps from the CPU generator right before the forwards (models/models.py:161)
        torch.manual_seed(99 + step)
        d["eps"] = (torch.empty(B, S).normal_(), torch.empty(B, S).normal_())
    return d


def product_step(srl, lm, losses, inp, validation):
    dev = srl.device
    if "vae" in losses:
        it = iter(inp["eps"])
        srl.model.model.eps_fn = lambda mu: next(it).to(mu.device)
    to = lambda t: None if t is None else t.to(dev)
    # as the learner's feed delivers them (SRL4robotics._toDevicePair): the two frames are the halves of ONE device buffer, which
    # is what lets the product take its default route (batched pair, reconstruction loss inside the last ConvTranspose)
    obs, next_obs = srl._toDevicePair(inp["obs"], inp["next_obs"])
    noisy = (None, None) if inp["noisy"][0] is None else srl._toDevicePair(inp["noisy"][0], inp["noisy"][1])
    loss = srl.trainStep(obs, next_obs, inp["actions"].view(-1, 1).to(dev), lm,
                         validation_mode=validation, noisy_obs=noisy[0], next_noisy_obs=noisy[1],
                         rewards_st=to(inp["rewards"]))
    rec = dict(zip(lm.names, lm.lossValues()))
    rec["total"] = float(loss.detach())
    return rec


def drive_product(losses, n_steps, val_steps=(), **cfg):
    """n_steps calls of SRL4robotics.trainStep on the fixture inputs; returns (learner, [per-step {name: value}])."""
    from losses.losses import LossManager
    srl = make_learner(losses, **cfg)
    lm = LossManager(srl.model, None)
    trace = [product_step(srl, lm, losses, step_inputs(losses, step, cfg.get("B", 2), cfg.get("S", 200)), step in val_steps)
             for step in range(n_steps)]
    torch.cuda.synchronize()
    return srl, trace


def _report(test, name, worst):
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open(os.path.join("gpurun_out", "trajectory_report.jsonl"), "a") as f:
            f.write(json.dumps({"test": test, "case": name, "worst": worst}, sort_keys=True) + "\n")
    except OSError:
        pass


@pytest.mark.parametrize("name", sorted(CASES))
def test_train_step_trajectory_follows_reference(name):
    cfg = dict(CASES[name])
    n_steps = cfg["n_steps"]
    srl, trace = drive_product(**cfg)
    # an optimiser step was taken for every training minibatch and none for a validation one
    assert srl.optimizer.steps() == n_steps - len(cfg.get("val_steps", ()))
    g = gu.load(name)
    fails = []
    names = [str(n) for n in g["trace/names"]]
    assert g["trace/values"].shape == (n_steps, len(names))
    worst_loss = 0.0
    for step, rec in enumerate(trace):
        assert sorted(rec.keys()) == sorted(names), (sorted(rec.keys()), names)
        for j, nm in enumerate(names):
            v = float(g["trace/values"][step, j])
            err = abs(rec[nm] - v) / max(abs(v), 1e-6)
            worst_loss = max(worst_loss, err)
            if not err <= LOSS_RTOL:
                fails.append("loss step %d %s: %.6g vs %.6g" % (step, nm, rec[nm], v))
    # end point (counters are asserted exactly inside endpoint_errors)
    sd = srl.model.state_dict()
    assert [str(k) for k in g["final/names"]] == list(sd.keys())
    srl.model.eval()
    obs, _, _ = gu.golden_inputs(int(g["eval_states/full"].shape[0]), 3, 6, seed=1234)
    with torch.no_grad():
        st = srl.model.getStates(torch.from_numpy(obs).to(srl.device)).double().cpu().numpy()
    worst, table = gu.endpoint_errors(sd, g, LR, n_steps, st)
    with open(os.path.join(gu.GOLDEN_DIR, "trajectory_spread.json")) as f:
        spreads = json.load(f)["cases"]
    big = lambda c: CASES[c].get("B", 2) >= 32  # a default-size minibatch is held to its own class's spread, not to B = 2's
    peers = [c for c in spreads if (CASES[c]["n_steps"] > 4) == (n_steps > 4) and big(c) == big(name)]
    spread = {kind: max(spreads[c][kind] for c in peers) for kind in worst}
    # (at the reference's default minibatch the product has been measured INSIDE the reference's own spread — eval-mode states 1.1 %
    # against 1.4 % — so that class is held to twice its spread instead of five times)
    factor = 2.0 if big(name) else SPREAD_FACTOR
    for kind, err in worst.items():
        tol = max(ENDPOINT_FLOOR[kind], factor * spread[kind])
        if not err <= tol:
            fails.append("end point %s: %.3e > %.3e (reference self-spread %.3e)" % (kind, err, tol, spread[kind]))
    worst["loss"] = worst_loss
    worst["worst_params"] = dict(sorted(table.items(), key=lambda kv: -kv[1])[:4])
    _report("free", name, worst)
    assert not fails, "\n".join(fails)


RIDE_CASES = ["trace_ae_b2", "trace_vae_b2", "trace_aeif_b2", "trace_split_dae_rfi_b4", "trace_val_aeif_b2", "trace_val_vae_b2",
              "trace_ae_l1l2_b2", "trace10_vae_b2",
              # the reference's default minibatch (configs[0]'s bs = 32): the tight per-step induction at the batch the reference runs
              "trace10_ae_b32"]


@pytest.mark.parametrize("name", RIDE_CASES)
def test_train_step_rides_along_the_oracle(name):
    """Per-step parity with the CPU oracle re-seeded from the product's state (see the module docstring, test 2)."""
    from collections import OrderedDict
    from losses.losses import LossManager
    from oracle import torch_twin as T
    cfg = dict(CASES[name])
    losses, n_steps, val_steps = cfg["losses"], cfg["n_steps"], cfg.get("val_steps", ())
    B, S = cfg.get("B", 2), cfg.get("S", 200)
    srl = make_learner(**cfg)
    lm = LossManager(srl.model, None)
    fp, opt = srl.flat_params, srl.optimizer
    pname = {id(p): n for n, p in srl.model.named_parameters()}
    slices = [(pname[id(p)], off, p.numel(), tuple(p.shape)) for p, off in zip(fp.params, fp.offsets)]
    taken = {}
    real_step = opt.step

    def step_spy(grad_scale=1.0):  # the gradient bucket exactly as Adam consumes it
        fp.deliver()
        taken["grad"] = fp.grad.detach().clone()
        return real_step(grad_scale)
    opt.step = step_spy
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    worst = {"loss": 0.0, "bn": 0.0, "grad_norm": 0.0, "grad_dir": 0.0, "adam": 0.0}
    fails = []

    def note(kind, err, tol, what):
        worst[kind] = max(worst[kind], err)
        if not err <= tol:
            fails.append("%s: %s err %.3e > %.1e" % (kind, what, err, tol))

    for step in range(n_steps):
        validation = step in val_steps
        inp = step_inputs(losses, step, B, S)
        before = OrderedDict((k, v.detach().cpu().clone()) for k, v in srl.model.state_dict().items())
        flat0, m0, v0, t0 = fp.flat.clone(), opt.m.clone(), opt.v.clone(), opt.steps()
        taken.clear()
        rec = product_step(srl, lm, losses, inp, validation)
        torch.cuda.synchronize()

        # ---- the oracle takes the same step from the same state
        sd = T.clone_state(before)
        ref = T.train_step(sd, losses, inp["obs"], inp["next_obs"], inp["actions"], eps=inp["eps"][0], next_eps=inp["eps"][1],
                           weights=cfg.get("weights"), split=cfg.get("split"), rewards=inp["rewards"],
                           l1_reg=cfg.get("l1_reg", 0.0), l2_reg=cfg.get("l2_reg", 0.0), noisy=inp["noisy"],
                           training=not validation)
        assert sorted(rec) == sorted(list(ref["losses"]) + ["total"])
        for nm, v in list(ref["losses"].items()) + [("total", ref["total"])]:
            note("loss", abs(rec[nm] - v) / max(abs(v), 1e-6), 1e-5, "step %d %s (%.7g vs %.7g)" % (step, nm, rec[nm], v))
        after = srl.model.state_dict()
        for k in before:
            if "running_" in k:
                r, got = sd[k].double(), after[k].double().cpu()
                note("bn", float((got - r).abs().max() / r.abs().max()), 1e-5, "step %d %s" % (step, k))
            elif "num_batches_tracked" in k:
                assert int(after[k]) == int(sd[k]), (step, k, int(after[k]), int(sd[k]))

        if validation:  # backward ran and was discarded: nothing that defines the trajectory may have moved
            assert "grad" not in taken and opt.steps() == t0
            assert torch.equal(fp.flat, flat0) and torch.equal(opt.m, m0) and torch.equal(opt.v, v0)
            for k in before:
                if "running_" in k:
                    assert torch.equal(after[k].cpu(), before[k]), (step, k)
            continue

        # ---- the gradient bucket Adam consumed vs the oracle's gradient
        assert opt.steps() == t0 + 1
        grad = taken["grad"].double().cpu()
        for nm, off, n, shape in slices:
            gref = ref["grads"].get(nm)
            g_got = grad[off:off + n]
            if gref is None:  # a head without a loss: torch skips it (grad None); the bucket must hold zeros
                assert float(g_got.abs().max()) == 0.0, (step, nm)
                continue
            gref = gref.double().reshape(-1)
            if nm in NOISE_BIASES:  # analytically zero: compare absolutely, on the scale of the layer's weight gradient
                scale = float(ref["grads"][nm.replace(".bias", ".weight")].abs().max())
                note("grad_dir", float((g_got - gref).abs().max()) / scale * 5e-2 / 1e-4, 5e-2, "step %d %s (noise bias)" % (step, nm))
                continue
            nr = float(gref.norm())
            note("grad_norm", abs(float(g_got.norm()) - nr) / max(nr, 1e-30), 5e-3, "step %d %s" % (step, nm))
            note("grad_dir", float((g_got - gref).norm()) / max(nr, 1e-30), 3e-2, "step %d %s" % (step, nm))

        # ---- Adam's update on THAT bucket, evaluated in fp64 (torch.optim.Adam defaults, models/learner.py:199)
        t = t0 + 1
        g64 = taken["grad"].double()
        m1 = 0.9 * m0.double() + 0.1 * g64
        v1 = 0.999 * v0.double() + 0.001 * g64 * g64
        expect = flat0.double() - (LR / (1 - 0.9 ** t)) * m1 / ((v1 / (1 - 0.999 ** t)).sqrt() + 1e-8)
        note("adam", float((fp.flat.double() - expect).abs().max()), 3e-7, "step %d parameter update" % step)
        assert float((opt.m.double() - m1).abs().max()) <= 1e-6 * max(float(m1.abs().max()), 1e-30)
        assert float((opt.v.double() - v1).abs().max()) <= 1e-6 * max(float(v1.abs().max()), 1e-30)
    _report("ride", name, worst)
    assert not fails, "\n".join(fails)


def test_staging_overflow_is_exercised():
    """l1 + l2 + two separately encoded frames (the two-call route, `_use_pair = False`) = four gradient contributions per regularised weight: the
    fourth finds no staging bucket (FlatParams.grad_buffer -> None) and must travel through autograd's own accumulation.
    The resulting step is checked against the oracle like any other."""
    from losses.losses import LossManager
    from oracle import torch_twin as T
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
        srl = make_learner(["autoencoder"], l1_reg=1e-5, l2_reg=1e-4)
        srl._use_pair = False
        before = {k: v.detach().cpu().clone() for k, v in srl.model.state_dict().items()}
        inp = step_inputs(["autoencoder"], 0, 2, 200)
        rec = product_step(srl, LossManager(srl.model, None), ["autoencoder"], inp, False)
        srl.flat_params.deliver()
        grad = srl.flat_params.grad.double().cpu()
    finally:
        optim.FlatParams.grad_buffer = orig
    assert calls["none"] > 0
    ref = T.train_step(T.clone_state(before), ["autoencoder"], inp["obs"], inp["next_obs"], inp["actions"], l1_reg=1e-5, l2_reg=1e-4)
    assert abs(rec["total"] - ref["total"]) <= 1e-5 * abs(ref["total"])
    pname = {id(p): n for n, p in srl.model.named_parameters()}
    for p, off in zip(srl.flat_params.params, srl.flat_params.offsets):
        nm = pname[id(p)]
        gref = ref["grads"].get(nm)
        if gref is None or nm in NOISE_BIASES:
            continue
        gref = gref.double().reshape(-1)
        assert float((grad[off:off + p.numel()] - gref).norm()) <= 3e-2 * float(gref.norm()), nm
