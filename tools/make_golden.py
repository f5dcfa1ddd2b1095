#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the UNMODIFIED reference (/root/reference) on CPU.

Runs only in the build container (the reference never travels to the GPU box).  The
reference is imported, never copied: four absent third-party modules are stubbed in
sys.modules (torchvision, cv2, termcolor, seaborn) exactly as SURVEY.md §8c describes.

Every fixture is DATA: inputs are regenerated from a seed by `golden_inputs()` (shared with
the tests through tests/golden_util.py), outputs are stored as sums, norms and strided
subsamples so the files stay small.

    python tools/make_golden.py            # writes tests/golden/*.npz
"""
from __future__ import print_function
import os
import sys
import types
import json

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, os.path.join(REPO, "tests"))


def _stub_modules():
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvm.resnet18 = lambda *a, **k: None
    tv.models = tvm
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm
    tc = types.ModuleType("termcolor")
    tc.colored = lambda s, *a, **k: s
    sys.modules["termcolor"] = tc
    sb = types.ModuleType("seaborn")
    sb.set = lambda *a, **k: None
    sys.modules["seaborn"] = sb
    sys.modules["cv2"] = types.ModuleType("cv2")
    import matplotlib
    matplotlib.use("Agg")


def import_reference():
    _stub_modules()
    sys.path.insert(0, REF)
    import torch as th
    th.set_num_threads(1)  # single-threaded: deterministic summation order
    import preprocessing.preprocess as ref_pre
    from models.modules import SRLModules
    import losses.losses as ref_losses
    return th, ref_pre, SRLModules, ref_losses


from golden_util import golden_inputs, golden_rewards, golden_noisy, tensor_digest  # noqa: E402


def digest_state_dict(sd):
    names, shapes, sums, abss = [], [], [], []
    for k, v in sd.items():
        names.append(k)
        shapes.append(list(v.shape))
        vv = v.double()
        sums.append(float(vv.sum()))
        abss.append(float(vv.abs().sum()))
    return dict(names=np.array(names), shapes=np.array([json.dumps(s) for s in shapes]),
                sums=np.array(sums), abss=np.array(abss))


def build(th, ref_pre, SRLModules, losses, S=200, A=6, C=3, seed=1, inverse="linear", split=None):
    ref_pre.N_CHANNELS = C
    np.random.seed(seed)
    th.manual_seed(seed)
    if split is not None:
        from models.modules import SRLModulesSplit  # the reference's
        return SRLModulesSplit(state_dim=S, action_dim=A, cuda=False, model_type="custom_cnn", losses=losses,
                               split_dimensions=split, inverse_model_type=inverse)
    return SRLModules(state_dim=S, action_dim=A, cuda=False, model_type="custom_cnn",
                      losses=losses, inverse_model_type=inverse)


def grads_digest(model, out, prefix="grad/"):
    for name, p in model.named_parameters():
        if p.grad is None:
            out[prefix + name + "/none"] = np.array(1)
            continue
        for k, v in tensor_digest(p.grad).items():
            out[prefix + name + "/" + k] = v


def bn_digest(model, out, prefix="bn/"):
    for k, v in model.state_dict().items():
        if "running_" in k or "num_batches" in k:
            out[prefix + k] = v.detach().double().numpy().copy()


def step_case(th, ref_pre, SRLModules, RL, losses, B, C=3, S=200, A=6, n_steps=1, lr=None,
              eps_seed=99, beta=1.0, inverse="linear", weights=None, split=None, l1_reg=0.0, l2_reg=0.0, val_steps=()):
    """One (or several) loop bodies of models/learner.py:373-497 driven on the reference classes.
    `val_steps`: steps run as validation minibatches (learner.py:362-364,487-497: eval mode, forward + backward, no
    optimizer step)."""
    model = build(th, ref_pre, SRLModules, [l for l in losses if l != "perceptual"] if split is None else losses, S=S, A=A,
                  C=C, inverse=inverse, split=split)
    denoiser = None
    if "perceptual" in losses:  # the frozen, eval-mode DAE of learner.py:317-326 (seed 7 stands in for "pre-trained")
        denoiser = build(th, ref_pre, SRLModules, ["dae"], S=S, A=A, C=C, seed=7)
        denoiser.eval()
        for param in denoiser.parameters():
            param.requires_grad = False
    w = {"forward": 1.0, "inverse": 2.0, "reward": 1.0, "autoencoder": 1.0, "dae": 1.0, "vae": 0.5e-6, "perceptual": 1e-6}
    if weights:
        w.update(weights)
    out = {}
    opt = None
    if lr is not None:
        opt = th.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=lr)
    history = {}
    lm = RL.LossManager(model, None)
    trace = []
    for step in range(n_steps):
        obs, next_obs, actions = golden_inputs(B, C, A, seed=1234 + step)
        obs, next_obs = th.from_numpy(obs), th.from_numpy(next_obs)
        act = th.from_numpy(actions).view(-1, 1)
        if step in val_steps:
            model.eval()
        else:
            model.train()
        if opt is not None:
            opt.zero_grad()
        lm.resetLosses()
        mu = logvar = None
        if l1_reg > 0:
            RL.l1Loss(lm.reg_params, l1_reg, lm)
        if l2_reg > 0:
            RL.l2Loss(lm.reg_params, l2_reg, lm)
        if "autoencoder" in losses:
            (states, dec), (next_states, next_dec) = model(obs), model(next_obs)
        elif "dae" in losses:
            noisy = th.from_numpy(golden_noisy(obs.numpy(), seed=1234 + step))
            next_noisy = th.from_numpy(golden_noisy(next_obs.numpy(), seed=4321 + step))
            (states, dec), (next_states, next_dec) = model(noisy), model(next_noisy)
        elif "vae" in losses:
            th.manual_seed(eps_seed + step)
            (dec, mu, logvar), (next_dec, next_mu, next_logvar) = model(obs), model(next_obs)
            states, next_states = model.getStates(obs), model.getStates(next_obs)
        else:
            states, next_states = model(obs), model(next_obs)
            dec = next_dec = None
        if "forward" in losses:
            pred = model.forwardModel(states, act)
            RL.forwardModelLoss(pred, next_states, weight=w["forward"], loss_manager=lm)
        if "inverse" in losses:
            logits = model.inverseModel(states, next_states)
            RL.inverseModelLoss(logits, act, weight=w["inverse"], loss_manager=lm)
        if "reward" in losses:
            _, rewards_st = golden_rewards(B, seed=1234 + step)
            rewards_pred = model.rewardModel(states, next_states)
            RL.rewardModelLoss(rewards_pred, th.from_numpy(rewards_st).long(), weight=w["reward"], loss_manager=lm)
        if "autoencoder" in losses or "dae" in losses:
            RL.autoEncoderLoss(obs, dec, next_obs, next_dec, weight=w["dae" if "dae" in losses else "autoencoder"],
                               loss_manager=lm)
        if "vae" in losses:
            RL.kullbackLeiblerLoss(mu, next_mu, logvar, next_logvar, loss_manager=lm, beta=beta)
            if denoiser is not None:
                (sd_real, _), (nsd_real, _) = denoiser(obs), denoiser(next_obs)
                (sd_pred, _), (nsd_pred, _) = denoiser(dec), denoiser(next_dec)
                RL.perceptualSimilarityLoss(sd_real, sd_pred, nsd_real, nsd_pred, weight=w["perceptual"], loss_manager=lm)
            else:
                RL.generationLoss(dec, next_dec, obs, next_obs, weight=w["vae"], loss_manager=lm)
        loss = lm.computeTotalLoss()
        loss.backward()
        rec = {n: float(l.item()) for n, l in zip(lm.names, lm.losses)}
        rec["total"] = float(loss.item())
        trace.append(rec)
        if step == 0:
            for n, v in rec.items():
                out["loss/" + n] = np.array(v)
            for k, v in tensor_digest(states).items():
                out["states/" + k] = v
            for k, v in tensor_digest(next_states).items():
                out["next_states/" + k] = v
            if dec is not None:
                for k, v in tensor_digest(dec).items():
                    out["decoded/" + k] = v
                for k, v in tensor_digest(next_dec).items():
                    out["next_decoded/" + k] = v
            if mu is not None:
                for k, v in tensor_digest(logvar).items():
                    out["logvar/" + k] = v
                for k, v in tensor_digest(next_logvar).items():
                    out["next_logvar/" + k] = v
            grads_digest(model, out)
            bn_digest(model, out)
        if opt is not None and step not in val_steps:
            opt.step()
    if n_steps > 1 or opt is not None:
        names = sorted(trace[0].keys())
        out["trace/names"] = np.array(names)
        out["trace/values"] = np.array([[t[n] for n in names] for t in trace])
        sd = digest_state_dict(model.state_dict())
        out["final/names"], out["final/sums"], out["final/abss"] = sd["names"], sd["sums"], sd["abss"]
        bn_digest(model, out, prefix="final_bn/")
    # eval-mode states on the (possibly updated) model: the "learned states" output (learner.py:67-88)
    model.eval()
    with th.no_grad():
        obs, _, _ = golden_inputs(B, C, A, seed=1234)
        st = model.getStates(th.from_numpy(obs))
    out["eval_states/full"] = st.double().numpy()
    return out


def layer_trace(th, ref_pre, SRLModules):
    """Per-layer forward digests of the AE (train mode, B=2) via forward hooks."""
    model = build(th, ref_pre, SRLModules, ["autoencoder"])
    obs, _, _ = golden_inputs(2, 3, 6, seed=1234)
    out = {}
    hooks = []

    def mk(name):
        def hook(_m, _i, o):
            for k, v in tensor_digest(o).items():
                out[name + "/" + k] = v
        return hook
    for name, mod in model.model.named_modules():
        if name.count(".") == 1 and (name.startswith("encoder_conv") or name.startswith("decoder_conv")
                                     or name.startswith("encoder_fc") or name.startswith("decoder_fc")):
            hooks.append(mod.register_forward_hook(mk(name)))
    model.train()
    model(th.from_numpy(obs))
    for h in hooks:
        h.remove()
    return out


def loss_kats(th, RL):
    """Known-answer vectors for the free loss functions (losses/losses.py:102-129,172-214,239-256)."""
    rs = np.random.RandomState(7)
    out = {}
    a = rs.randn(4, 3, 8, 8).astype(np.float32)
    b = rs.randn(4, 3, 8, 8).astype(np.float32)
    c = rs.randn(4, 3, 8, 8).astype(np.float32)
    d = rs.randn(4, 3, 8, 8).astype(np.float32)
    mu, nmu = rs.randn(4, 10).astype(np.float32), rs.randn(4, 10).astype(np.float32)
    lv, nlv = (0.3 * rs.randn(4, 10)).astype(np.float32), (0.3 * rs.randn(4, 10)).astype(np.float32)
    logits = rs.randn(4, 6).astype(np.float32)
    act = rs.randint(0, 6, (4, 1)).astype(np.int64)
    for k, v in dict(a=a, b=b, c=c, d=d, mu=mu, nmu=nmu, lv=lv, nlv=nlv, logits=logits, act=act).items():
        out["in/" + k] = v
    T = th.from_numpy

    class M(th.nn.Module):
        pass
    lm = RL.LossManager(M(), {})
    lm.loss_history = __import__("collections").defaultdict(list)
    out["reconstruction"] = np.array(RL.reconstructionLoss(T(a), T(b)).item())
    out["autoencoder_w1"] = np.array(RL.autoEncoderLoss(T(a), T(b), T(c), T(d), 1.0, lm).item())
    out["generation_w"] = np.array(RL.generationLoss(T(b), T(d), T(a), T(c), 0.5e-6, lm).item())
    out["kl_beta2"] = np.array(RL.kullbackLeiblerLoss(T(mu), T(nmu), T(lv), T(nlv), lm, beta=2.0).item())
    out["forward_w1"] = np.array(RL.forwardModelLoss(T(mu), T(nmu), 1.0, lm).item())
    out["inverse_w2"] = np.array(RL.inverseModelLoss(T(logits), T(act), 2.0, lm).item())
    sa, sp, sn = rs.randn(5, 7).astype(np.float32), rs.randn(5, 7).astype(np.float32), rs.randn(5, 7).astype(np.float32)
    out["in/tri_s"], out["in/tri_p"], out["in/tri_n"] = sa, sp, sn
    lm2 = RL.LossManager(M(), {})
    out["triplet_w1"] = np.array(RL.tripletLoss(T(sa), T(sp), T(sn), 1.0, lm2, alpha=0.2).item())
    out["total"] = np.array(lm.computeTotalLoss().item())
    lm.updateLossHistory()
    lm.updateLossHistory()
    out["history/names"] = np.array(list(lm.loss_history.keys()))
    out["history/values"] = np.array([lm.loss_history[k][-1] for k in lm.loss_history.keys()])
    out["names"] = np.array(lm.names)
    out["weights"] = np.array(lm.weights, dtype=np.float64)
    return out


def head_kats(th, ref_pre, SRLModules):
    """forwardModel / inverseModel (linear + mlp) outputs (models/forward_inverse.py:21-31,62-70)."""
    out = {}
    rs = np.random.RandomState(11)
    s = rs.randn(5, 200).astype(np.float32)
    ns = rs.randn(5, 200).astype(np.float32)
    act = rs.randint(0, 6, (5, 1)).astype(np.int64)
    out["in/s"], out["in/ns"], out["in/act"] = s, ns, act
    for inv in ("linear", "mlp"):
        m = build(th, ref_pre, SRLModules, ["autoencoder", "inverse", "forward"], inverse=inv)
        out[inv + "/forward"] = m.forwardModel(th.from_numpy(s), th.from_numpy(act)).detach().numpy()
        out[inv + "/inverse"] = m.inverseModel(th.from_numpy(s), th.from_numpy(ns)).detach().numpy()
    return out


def detach_kats(th):
    """SRLModulesSplit.detachSplit on an all-ones state for a list of split configurations: the kept-column mask per
    (configuration, index).  Uses the reference's method unbound on a minimal stand-in object (it only reads
    self.split_dimensions)."""
    from collections import OrderedDict as OD
    from models.modules import SRLModulesSplit
    configs = [OD([("dae", 20), ("reward", -1), ("forward", 60), ("inverse", 20)]),
               OD([("vae", 150), ("inverse", 50), ("forward", -1)]),
               OD([("autoencoder", 120), ("reward", 80), ("inverse", -1)]),
               OD([("autoencoder", 50), ("inverse", -1), ("forward", -1)]),
               OD([("autoencoder", 30), ("inverse", -1), ("forward", -1), ("reward", 20)]),
               OD([("inverse", 10), ("forward", 10), ("reward", 10)])]
    out = {"n_configs": np.array(len(configs))}

    class Holder(object):
        pass
    for ci, cfg in enumerate(configs):
        S = sum(v for v in cfg.values() if v > 0)
        h = Holder()
        h.split_dimensions = cfg
        out["cfg%d/keys" % ci] = np.array(list(cfg.keys()))
        out["cfg%d/dims" % ci] = np.array(list(cfg.values()))
        x = th.ones(2, S)
        for index in list(cfg.keys()) + ["autoencoder", "vae", "absent"]:
            try:
                y = SRLModulesSplit.detachSplit(h, x, index)
                out["cfg%d/mask/%s" % (ci, index)] = y[0].numpy().astype(np.int8)
            except Exception as e:  # e.g. torch.cat of an empty list when nothing is kept and nothing is zeroed
                out["cfg%d/error/%s" % (ci, index)] = np.array(type(e).__name__)
    return out


def _install_cv2_shim():
    """PIL-backed stand-in for the five cv2 symbols the reference loader touches (data_loader.py:48-51,212,238,247):
    imread -> BGR ndarray or None, resize (identity for 224x224 sources, the only case used), cvtColor(BGR2RGB),
    INTER_AREA, COLOR_BGR2RGB.  Everything downstream of the decoded uint8 frame is the reference's own code."""
    from PIL import Image
    cv2 = sys.modules["cv2"]

    def imread(path):
        try:
            with Image.open(path) as f:
                return np.ascontiguousarray(np.asarray(f.convert("RGB"))[..., ::-1])
        except (IOError, OSError):
            return None

    def resize(im, size, interpolation=None):
        assert (im.shape[1], im.shape[0]) == tuple(size), "shim: only already-sized frames"
        return im

    cv2.imread, cv2.resize = imread, resize
    cv2.cvtColor = lambda im, code: np.ascontiguousarray(im[..., ::-1])
    cv2.INTER_AREA, cv2.COLOR_BGR2RGB = 3, 4


def loop_case(th, losses, n_epochs=2, bs=8, S=12, seed=3, lr=1e-4, n_episodes=4, ep_len=26, **ctor):
    """The UNMODIFIED SRL4robotics.learn() (models/learner.py:259-579: forked loader process, queue, train/validation
    split, best-model checkpoint, state prediction) on the tiny generated dataset of tests/dataset_util.py."""
    import tempfile
    import shutil
    from dataset_util import make_dataset
    _install_cv2_shim()
    import models.learner as RLn
    tmp = tempfile.mkdtemp(prefix="srlz_loop_")
    cwd = os.getcwd()
    try:
        name, paths, actions, rewards, starts = make_dataset(tmp, n_episodes=n_episodes, ep_len=ep_len)
        os.chdir(tmp)
        os.makedirs("logs/run", exist_ok=True)
        RLn.DISPLAY_PLOTS, RLn.N_EPOCHS, RLn.BATCH_SIZE, RLn.VALIDATION_SIZE = False, n_epochs, bs, 0.2
        srl = RLn.SRL4robotics(S, model_type="custom_cnn", seed=seed, learning_rate=lr, cuda=False, losses=losses,
                               n_actions=6, log_folder="logs/run", **ctor)
        loss_history, states, pairs = srl.learn(paths, actions, rewards, starts)
        out = {"states/full": np.asarray(states, dtype=np.float64),
               "pairs/names": np.array([p[0] for p in pairs]), "pairs/weights": np.array([float(p[1]) for p in pairs]),
               "history/names": np.array(sorted(loss_history.keys())),
               "history/values": np.array([loss_history[k] for k in sorted(loss_history.keys())], dtype=np.float64)}
        sd = digest_state_dict(th.load("logs/run/srl_model.pth"))
        out["final/names"], out["final/sums"], out["final/abss"] = sd["names"], sd["sums"], sd["abss"]
        out["config"] = np.array(json.dumps(dict(losses=losses, n_epochs=n_epochs, bs=bs, S=S, seed=seed, lr=lr,
                                                 n_episodes=n_episodes, ep_len=ep_len, ctor=ctor)))
        return out
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


# whole-loop cases run in a FRESH interpreter each: learn() forks its loader processes, and a child forked from a parent
# that has already run OpenMP-parallel torch code dead-locks in its first parallel region
LOOP_CASES = {"loop_aeif": dict(losses=["autoencoder", "inverse", "forward"]),
              "loop_ae_reward": dict(losses=["autoencoder", "reward"], n_epochs=1, seed=5, l2_reg=1e-4)}


def run_loop_child(name):
    import subprocess
    import tempfile
    tmp = tempfile.mktemp(suffix=".npz")
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--loop-child", name, tmp], timeout=1500)
    with np.load(tmp, allow_pickle=False) as z:
        d = {k: z[k] for k in z.files}
    os.remove(tmp)
    return d


def loop_child(name, path):
    _stub_modules()
    sys.path.insert(0, REF)
    import torch as th
    th.set_num_threads(1)  # before anything runs: deterministic summation order, and nothing for a forked child to trip over
    np.savez_compressed(path, **loop_case(th, **LOOP_CASES[name]))


def main():
    os.makedirs(OUT, exist_ok=True)
    th, ref_pre, SRLModules, RL = import_reference()

    only = [a for a in sys.argv[1:] if not a.startswith("-")]

    def save(name, fn):
        if only and not any(name.startswith(o) for o in only):
            return
        d = fn()
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **d)
        print("wrote %-28s %6.1f KB  (%d arrays)" % (name + ".npz", os.path.getsize(path) / 1024.0, len(d)))

    # (1) init KATs: same seed + same constructor order => identical parameters
    for tag, losses, C in (("ae_c3", ["autoencoder"], 3), ("vae_c3", ["vae"], 3), ("ae_c6", ["autoencoder"], 6),
                           ("cnn_c3", ["inverse"], 3)):
        save("init_" + tag, lambda: digest_state_dict(build(th, ref_pre, SRLModules, losses, C=C).state_dict()))
    # (2) single train-mode steps
    save("step_ae_b2", lambda: step_case(th, ref_pre, SRLModules, RL, ["autoencoder"], B=2))
    save("step_ae_b4", lambda: step_case(th, ref_pre, SRLModules, RL, ["autoencoder"], B=4))
    save("step_vae_b2", lambda: step_case(th, ref_pre, SRLModules, RL, ["vae"], B=2))
    save("step_vae_b4", lambda: step_case(th, ref_pre, SRLModules, RL, ["vae"], B=4))
    save("step_aeif_b2", lambda: step_case(th, ref_pre, SRLModules, RL, ["autoencoder", "inverse", "forward"], B=2))
    save("step_aeif_mlp_b2", lambda: step_case(th, ref_pre, SRLModules, RL, ["autoencoder", "inverse", "forward"], B=2,
                                       inverse="mlp"))
    save("step_ae_c6_b2", lambda: step_case(th, ref_pre, SRLModules, RL, ["autoencoder"], B=2, C=6))
    save("step_vae_c6_b2", lambda: step_case(th, ref_pre, SRLModules, RL, ["vae"], B=2, C=6))
    save("step_cnn_if_b2", lambda: step_case(th, ref_pre, SRLModules, RL, ["inverse", "forward"], B=2))
    # (3) short optimisation traces (Adam, lr 1e-4)
    save("trace_ae_b2", lambda: step_case(th, ref_pre, SRLModules, RL, ["autoencoder"], B=2, n_steps=3, lr=1e-4))
    save("trace_vae_b2", lambda: step_case(th, ref_pre, SRLModules, RL, ["vae"], B=2, n_steps=3, lr=1e-4))
    save("trace_aeif_b2", lambda: step_case(th, ref_pre, SRLModules, RL, ["autoencoder", "inverse", "forward"], B=2,
                                    n_steps=3, lr=1e-4))
    # (3b) §8f-2/3: split representations (SRLModulesSplit), reward head + loss, l1/l2 regularisers, DAE inputs.
    # The first case is the reference's own stacked-model test configuration (tests/test_modules.py:8-19) at B=4.
    from collections import OrderedDict as OD
    stacked = OD([("dae", 20), ("reward", -1), ("forward", 60), ("inverse", 20)])
    stacked_w = {"dae": 1.0, "reward": 1.0, "forward": 1.0, "inverse": 5.0}
    save("step_split_dae_rfi_b4", lambda: step_case(th, ref_pre, SRLModules, RL, list(stacked.keys()), B=4, S=100, inverse="mlp",
                                            weights=stacked_w, split=stacked, l2_reg=0.0001))
    save("trace_split_dae_rfi_b4", lambda: step_case(th, ref_pre, SRLModules, RL, list(stacked.keys()), B=4, S=100, inverse="mlp",
                                             weights=stacked_w, split=stacked, l2_reg=0.0001, n_steps=3, lr=1e-4))
    vsplit = OD([("vae", 150), ("inverse", 50), ("forward", -1)])
    save("step_split_vae_if_b2", lambda: step_case(th, ref_pre, SRLModules, RL, list(vsplit.keys()), B=2, split=vsplit))
    asplit = OD([("autoencoder", 120), ("reward", 80), ("inverse", -1)])
    save("step_split_ae_ri_b2", lambda: step_case(th, ref_pre, SRLModules, RL, list(asplit.keys()), B=2, split=asplit))
    save("step_ae_reward_l1_b2", lambda: step_case(th, ref_pre, SRLModules, RL, ["autoencoder", "reward"], B=2, l1_reg=1e-5))
    save("step_dae_b2", lambda: step_case(th, ref_pre, SRLModules, RL, ["dae"], B=2))
    save("step_vae_perceptual_b2", lambda: step_case(th, ref_pre, SRLModules, RL, ["vae", "perceptual"], B=2,
                                             weights={"perceptual": 1.0}))
    save("detach_kats", lambda: detach_kats(th))
    # (4) per-layer forward digests, loss KATs, head KATs
    save("layers_ae_b2", lambda: layer_trace(th, ref_pre, SRLModules))
    save("loss_kats", lambda: loss_kats(th, RL))
    save("head_kats", lambda: head_kats(th, ref_pre, SRLModules))
    # (5) round 2: longer trajectories, a validation minibatch inside a trajectory, the whole learn() loop
    save("trace10_ae_b2", lambda: step_case(th, ref_pre, SRLModules, RL, ["autoencoder"], B=2, n_steps=10, lr=1e-4))
    save("trace10_vae_b2", lambda: step_case(th, ref_pre, SRLModules, RL, ["vae"], B=2, n_steps=10, lr=1e-4))
    save("trace_val_aeif_b2", lambda: step_case(th, ref_pre, SRLModules, RL, ["autoencoder", "inverse", "forward"], B=2,
                                                n_steps=4, lr=1e-4, val_steps=(1,)))
    save("trace_val_vae_b2", lambda: step_case(th, ref_pre, SRLModules, RL, ["vae"], B=2, n_steps=4, lr=1e-4,
                                               val_steps=(2,)))
    save("trace_ae_l1l2_b2", lambda: step_case(th, ref_pre, SRLModules, RL, ["autoencoder"], B=2, n_steps=3, lr=1e-4,
                                               l1_reg=1e-5, l2_reg=1e-4))
    # (6) round 3: a trajectory at the reference's DEFAULT minibatch (bs = 32, BASELINE.json configs[0]): with 16x the samples per
    # gradient, near-zero gradient elements (the ones Adam turns into +-lr steps of arbitrary sign) are far rarer than at B = 2,
    # so a free-running end point can be held much tighter (tests/test_trajectory_gpu.py, tools/measure_spread.py)
    save("trace10_ae_b32", lambda: step_case(th, ref_pre, SRLModules, RL, ["autoencoder"], B=32, n_steps=10, lr=1e-4))
    for lname in LOOP_CASES:
        save(lname, lambda: run_loop_child(lname))


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "--loop-child":
        loop_child(sys.argv[2], sys.argv[3])
    else:
        main()
