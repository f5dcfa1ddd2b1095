"""Shared helpers for the golden fixtures (used by tools/make_golden.py and by the tests).

Inputs are never stored: they are regenerated from a seed here, so a fixture holds only
expected outputs.  `tensor_digest` is the reduced form an output tensor is stored in.
"""
import os
import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SUB = 97  # stride of the flat subsample kept for big tensors

_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def synthetic_obs(B, C, seed):
    """uint8 noise image -> /255 -> ImageNet mean/std -> reference tensor layout [B,C,W,H].

    Mirrors what preprocessing/data_loader.py:38-65,255 + preprocessing/utils.py:20-32 do to a
    decoded frame (SURVEY §8d "Configs 2-4"): per RGB triple normalisation, then
    transpose(0,3,2,1).  Returns (obs, next_obs) float32.
    """
    rs = np.random.RandomState(seed)
    raw = rs.randint(0, 256, (2, B, 224, 224, C)).astype(np.float32) / np.float32(255.0)
    reps = C // 3
    raw = (raw - np.tile(_MEAN, reps)) / np.tile(_STD, reps)
    x = np.ascontiguousarray(raw.transpose(0, 1, 4, 3, 2)).astype(np.float32)
    return x[0], x[1]


def golden_inputs(B, C, A, seed=1234):
    obs, next_obs = synthetic_obs(B, C, seed)
    actions = np.random.RandomState(seed + 77).randint(0, A, (B,)).astype(np.int64)
    return obs, next_obs, actions


def golden_rewards(B, seed=1234):
    """Rewards as the datasets store them (-1 / 0 / 1) and as the loop feeds them to the reward loss (-1 -> 0, int64;
    reference models/learner.py:439-442)."""
    raw = np.random.RandomState(seed + 33).randint(-1, 2, (B,)).astype(np.int64)
    st = raw.copy()
    st[st == -1] = 0
    return raw, st


def golden_noisy(obs, seed=1234):
    """A deterministic occluded copy of a batch (what the DAE sees): one zeroed rectangle per image."""
    rs = np.random.RandomState(seed + 55)
    out = obs.copy()
    for i in range(out.shape[0]):
        h1, w1 = rs.randint(0, 112, 2)
        h2, w2 = h1 + rs.randint(20, 112), w1 + rs.randint(20, 112)
        out[i, :, h1:h2, w1:w2] = 0.0
    return out


def ext_cases():
    """§8f-2/3 fixtures (tools/make_golden.py section 3b): name -> configuration of the step.  The first one is the
    reference's own stacked-model test configuration (tests/test_modules.py:8-19) at B=4."""
    from collections import OrderedDict as OD
    stacked = OD([("dae", 20), ("reward", -1), ("forward", 60), ("inverse", 20)])
    return {
        "step_split_dae_rfi_b4": dict(losses=list(stacked.keys()), B=4, S=100, inverse="mlp", split=stacked, l2_reg=0.0001,
                                      weights={"dae": 1.0, "reward": 1.0, "forward": 1.0, "inverse": 5.0}),
        "step_split_vae_if_b2": dict(losses=["vae", "inverse", "forward"], B=2,
                                     split=OD([("vae", 150), ("inverse", 50), ("forward", -1)])),
        "step_split_ae_ri_b2": dict(losses=["autoencoder", "reward", "inverse"], B=2,
                                    split=OD([("autoencoder", 120), ("reward", 80), ("inverse", -1)])),
        "step_ae_reward_l1_b2": dict(losses=["autoencoder", "reward"], B=2, l1_reg=1e-5),
        "step_dae_b2": dict(losses=["dae"], B=2),
        # VAE trained through a frozen denoiser (DARLA's perceptual similarity); the denoiser is the
        # seed-7 initialisation of SRLModules(losses=["dae"]) in eval mode
        "step_vae_perceptual_b2": dict(losses=["vae", "perceptual"], B=2, weights={"perceptual": 1.0}, dae_seed=7),
    }


def ext_defaults(cfg):
    out = dict(S=200, inverse="linear", split=None, l1_reg=0.0, l2_reg=0.0, weights=None, dae_seed=None)
    out.update(cfg)
    return out


def tensor_digest(t):
    """sum, abs-sum, L2 norm (float64) and a strided flat subsample of a tensor/ndarray."""
    if hasattr(t, "detach"):
        t = t.detach().cpu().double().numpy()
    a = np.asarray(t, dtype=np.float64).reshape(-1)
    return {
        "sum": np.array(a.sum()),
        "abs": np.array(np.abs(a).sum()),
        "l2": np.array(np.sqrt((a * a).sum())),
        "sub": a[::SUB].copy() if a.size > 4096 else a.copy(),
    }


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)


def check_digest(t, g, prefix, rtol=1e-4, atol_scale=1e-5):
    """Assert tensor `t` matches the stored digest g[prefix+/...]; returns worst relative error."""
    d = tensor_digest(t)
    ref_sub = g[prefix + "/sub"]
    scale = max(float(np.abs(ref_sub).max()), 1e-30)
    err_sub = float(np.abs(d["sub"] - ref_sub).max()) / scale
    n = max(d["sub"].size, 1)
    l2 = float(g[prefix + "/l2"])
    err_l2 = abs(float(d["l2"]) - l2) / max(l2, 1e-30)
    ab = float(g[prefix + "/abs"])
    err_sum = abs(float(d["sum"]) - float(g[prefix + "/sum"])) / max(ab, 1e-30)
    worst = max(err_sub, err_l2, err_sum)
    assert err_sub <= rtol, "%s: subsample rel err %.3e > %.1e" % (prefix, err_sub, rtol)
    assert err_l2 <= rtol, "%s: l2 rel err %.3e" % (prefix, err_l2)
    assert err_sum <= rtol, "%s: sum err (rel. to abs-sum) %.3e" % (prefix, err_sum)
    return worst


NOISE_BIASES = tuple("model.decoder_conv.%d.bias" % i for i in (0, 3, 6, 9))


def endpoint_errors(sd, g, lr, n_steps, eval_states):
    """Errors of a trajectory END POINT (`sd`: name -> tensor after the last step, `eval_states`: eval-mode states of that
    model on the fixture's first batch) against a trace fixture `g` -> {"param", "noise_bias", "bn", "eval_states"} plus the
    per-tensor table.  Parameters are measured as max(|sum - ref|, |abs-sum - ref|) / (abs-sum + lr*n_steps*numel): the
    second term is the distance Adam can travel in n_steps, which is the natural scale for parameters that start at
    zero (BatchNorm biases).  Counters (num_batches_tracked) must match exactly and are asserted here."""
    import numpy as _np
    worst = {"param": 0.0, "noise_bias": 0.0, "bn": 0.0}
    table = {}
    for k, ref_sum, ref_abs in zip(g["final/names"], g["final/sums"], g["final/abss"]):
        k = str(k)
        v = sd[k].detach().double().cpu()
        if "num_batches_tracked" in k:
            assert int(v) == int(ref_sum), (k, int(v), int(ref_sum))
            continue
        err = max(abs(float(v.sum()) - ref_sum), abs(float(v.abs().sum()) - ref_abs))
        err /= (ref_abs + lr * n_steps * v.numel())
        kind = "noise_bias" if k in NOISE_BIASES else ("bn" if "running_" in k else "param")
        worst[kind] = max(worst[kind], err)
        table[k] = err
    ref = g["eval_states/full"]
    worst["eval_states"] = float(_np.abs(_np.asarray(eval_states, dtype=_np.float64) - ref).max() / _np.abs(ref).max())
    return worst, table


def pins_from_observed(observed, B):
    """(decisions of obs, decisions of next_obs) of ONE batched step: pool argmax (flat H*W index per window) + positivity of the
    pooled value, decoder ReLU masks — the `pins` format of oracle.torch_twin.train_step.  The batch is [obs ; next_obs] (N = 2 B)."""
    import torch
    from srlz import ops
    full = {}
    for name, pad in (("encoder_conv.3", 1), ("encoder_conv.7", 0), ("encoder_conv.11", 0)):
        pooled, saved = observed[name]
        y, _bnp, arg = saved[:3]
        n, h, w, _ = y.shape
        a = arg.long().cpu().permute(0, 3, 1, 2)  # [n, c, hp, wp]: window index ky * 3 + kx
        hp, wp = a.shape[2], a.shape[3]
        py, px = torch.arange(hp).view(1, 1, hp, 1), torch.arange(wp).view(1, 1, 1, wp)
        idx = (py * 2 - pad + a // 3) * w + (px * 2 - pad + a % 3)
        pooled = pooled.cpu()
        if pooled.shape != a.shape:
            pooled = pooled.permute(0, 3, 1, 2)
        full[name] = (idx, pooled > 0)
    for node, relu in ((3, 2), (6, 5), (9, 8), (12, 11)):
        key = "decoder_conv.%d" % node
        if key not in observed:
            continue
        y_prev, bnp = observed[key][1][:2]
        groups = bnp.numel() // 256
        per = y_prev.shape[0] // groups
        act = torch.cat([ops.bn_relu_materialise(y_prev[g * per:(g + 1) * per].contiguous(), bnp[256 * g:256 * (g + 1)].contiguous())
                         for g in range(groups)], 0)
        full["decoder_conv.%d" % relu] = act.cpu().permute(0, 3, 1, 2) > 0
    halves = ({}, {})
    for k, v in full.items():
        for i in range(2):
            sl = slice(i * B, (i + 1) * B)
            halves[i][k] = (v[0][sl], v[1][sl]) if isinstance(v, tuple) else v[sl]
    return halves
