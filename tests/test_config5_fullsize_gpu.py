"""BASELINE.json configs[4] at its FULL per-GPU size (bs = 128), both reference-valid halves (the combined flag
set `vae triplet` crashes in the reference itself):

  * `--multi-view --losses vae`     6-channel CNNVAE (reference models/vae.py:43-75, train.py:116-123), and
  * `--multi-view --losses triplet` 9-channel triplets through EmbeddingNet: six frozen ResNet-18 passes per step (reference
                                    models/triplet.py:6-39, models/learner.py:383-391),

through the product's own loop body (SRL4robotics.trainStep) against the CPU oracle at the same size: every loss term, a sample
of the states, the BatchNorm running statistics and counters after the step (VAE: four momentum updates of the three encoder
layers; triplet: six train-mode passes over all 20 BatchNorm layers of the trunk, i.e. `convN_fwd_kernel` on 128 x 512-channel
7x7 maps and the chunked BatchNorm finalize AT SIZE), and bit-for-bit determinism of a second run.  Plus the pre-trained-trunk
path: a state_dict with torchvision's resnet18 keys loaded through SRLZ_RESNET18_WEIGHTS (reference models/triplet.py:16
`resnet18(pretrained=True)`).

PARITY UNPINNED for the ResNet-18 trunk (torchvision is absent: the oracle restates its published definition, oracle/torch_twin.py).
"""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu
B = 128


def _threads():
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))


def _learner(losses, channels, S=200, seed=11):
    import models.learner as learner
    import preprocessing.preprocess as pre
    pre.N_CHANNELS = channels
    learner.BATCH_SIZE = B
    return learner.SRL4robotics(S, model_type="custom_cnn", seed=seed, learning_rate=1e-4, cuda=True, losses=losses, n_actions=6,
                                log_folder="/tmp", multi_view=True)


def _spy_step(srl):
    taken = {}
    orig = srl.optimizer.step

    def step_spy(grad_scale=1.0):  # the gradient bucket as Adam sees it
        srl.flat_params.deliver()
        taken["grad"] = srl.flat_params.grad.clone()
        return orig(grad_scale)
    srl.optimizer.step = step_spy
    return taken


def _check_buffers(sd_ref, sd_got, tol=1e-4):
    for k in sd_ref:
        if "running_" in k:
            r, g = sd_ref[k].double(), sd_got[k].double().cpu()
            assert float((g - r).abs().max()) <= tol * max(float(r.abs().max()), 1e-30), k
        elif "num_batches_tracked" in k:
            assert int(sd_got[k]) == int(sd_ref[k]), (k, int(sd_got[k]), int(sd_ref[k]))


def test_full_size_multi_view_vae():
    """configs[4], `vae` half: two stacked camera views (C = 6), bs = 128."""
    import torch.nn.functional as F
    from losses.losses import LossManager
    from oracle import torch_twin as T
    _threads()
    obs_np, next_np, act_np = gu.golden_inputs(B, 6, 6, seed=606)
    obs, nxt, act = torch.from_numpy(obs_np), torch.from_numpy(next_np), torch.from_numpy(act_np)
    torch.manual_seed(6)
    eps, next_eps = torch.randn(B, 200), torch.randn(B, 200)

    def run():
        srl = _learner(["vae"], 6)
        init = OrderedDict((k, v.detach().cpu().clone()) for k, v in srl.model.state_dict().items())
        it = iter((eps, next_eps))
        srl.model.model.eps_fn = lambda mu: next(it).to(mu.device)
        lm = LossManager(srl.model, None)
        taken = _spy_step(srl)
        o, no = srl._toDevicePair(obs, nxt)  # the learner's feed: the product's default route (batched pair, fused loss)
        total = srl.trainStep(o, no, act.view(-1, 1).cuda(), lm)
        vals = dict(zip(lm.names, lm.lossValues()))
        torch.cuda.synchronize()
        return srl, init, vals, float(total.detach()), taken["grad"]

    srl, init, vals, total, grad = run()
    assert tuple(init["model.encoder_conv.0.weight"].shape) == (64, 6, 7, 7)
    assert tuple(init["model.decoder_conv.12.weight"].shape) == (64, 6, 4, 4)
    sd = T.clone_state(init, requires_grad=False)
    with torch.no_grad():
        dec, mu, logvar = T.vae_forward(sd, obs, True, eps)
        ndec, nmu, nlogvar = T.vae_forward(sd, nxt, True, next_eps)
        T.vae_encode(sd, obs, True)  # the getStates quirk (reference learner.py:402): two more train-mode encoder passes
        T.vae_encode(sd, nxt, True)
        ref = {"kl_loss": float(T.kl_loss(mu, logvar) + T.kl_loss(nmu, nlogvar)),
               "generation_loss": float(F.mse_loss(dec, obs, reduction="sum") + F.mse_loss(ndec, nxt, reduction="sum"))}
    assert sorted(vals) == sorted(ref)
    for k in ref:
        assert abs(vals[k] - ref[k]) <= 1e-4 * abs(ref[k]), (k, vals[k], ref[k])
    assert abs(total - (ref["kl_loss"] + 0.5e-6 * ref["generation_loss"])) <= 1e-4 * abs(total)
    _check_buffers(sd, srl.model.state_dict())
    # eval-mode states of a sample of the batch on the UPDATED model against the oracle on the same parameters
    srl.model.eval()
    sel = torch.tensor([0, 63, 127])
    with torch.no_grad():
        st = srl.model.getStates(obs[sel].cuda()).cpu()
    upd = T.clone_state(OrderedDict((k, v.detach().cpu()) for k, v in srl.model.state_dict().items()), requires_grad=False)
    ref_st = T.get_states(upd, obs[sel], "vae")
    assert float((st - ref_st).abs().max()) <= 1e-4 * float(ref_st.abs().max())
    # determinism
    srl2, _, vals2, total2, grad2 = run()
    assert vals2 == vals and total2 == total and torch.equal(grad, grad2)
    assert torch.isfinite(grad).all() and float(grad.abs().max()) > 0


def _views(n, seed):
    frames = [gu.synthetic_obs(n, 3, seed + i) for i in range(3)]
    return (torch.from_numpy(np.concatenate([f[0] for f in frames], axis=1)),
            torch.from_numpy(np.concatenate([f[1] for f in frames], axis=1)))


@pytest.mark.timeout(900)
def test_full_size_triplet():
    """configs[4], `triplet` half: anchor / positive / negative views (C = 9), bs = 128, six trunk passes of 128 images."""
    from losses.losses import LossManager
    from oracle import torch_twin as T
    _threads()
    obs, nxt = _views(B, 909)
    act = torch.from_numpy(np.random.RandomState(9).randint(0, 6, (B,)).astype(np.int64))
    S = 200

    def run():
        srl = _learner(["triplet"], 9, S=S, seed=4)
        init = OrderedDict((k, v.detach().cpu().clone()) for k, v in srl.model.state_dict().items())
        lm = LossManager(srl.model, None)
        taken = _spy_step(srl)
        total = srl.trainStep(obs.cuda(), nxt.cuda(), act.view(-1, 1).cuda(), lm)
        vals = dict(zip(lm.names, lm.lossValues()))
        torch.cuda.synchronize()
        return srl, init, vals, float(total.detach()), taken["grad"]

    srl, init, vals, total, grad = run()
    sd = T.clone_state(init)
    ref = T.train_step(sd, ["triplet"], obs, nxt, act, training=True)  # six train-mode trunk passes on the host cores
    assert sorted(vals) == sorted(ref["losses"])
    for k, v in ref["losses"].items():
        assert abs(vals[k] - v) <= 1e-4 * max(abs(v), 1e-6), (k, vals[k], v)
    assert abs(total - ref["total"]) <= 1e-4 * max(abs(ref["total"]), 1e-6)
    got = srl.model.state_dict()
    n_bn = 0
    for k in sd:
        if "running_mean" in k and "conv_layers" in k:
            n_bn += 1
        if "num_batches_tracked" in k and "conv_layers" in k:
            assert int(got[k]) == 6, (k, int(got[k]))  # three views x two frames, train mode (reference learner.py:365,383-391)
    assert n_bn == 20
    _check_buffers(sd, got)
    # the head's gradients (the trunk is frozen: not in the bucket)
    pname = {id(p): n for n, p in srl.model.named_parameters()}
    g64 = grad.double().cpu()
    for p, off in zip(srl.flat_params.params, srl.flat_params.offsets):
        nm = pname[id(p)]
        assert not nm.startswith("model.conv_layers.layer")
        gref = ref["grads"].get(nm)
        if gref is None:
            assert float(g64[off:off + p.numel()].abs().max()) == 0.0, nm
            continue
        gref = gref.double().reshape(-1)
        scale = float(gref.norm())
        if nm == "model.fc.1.bias":
            # analytically ZERO: the triplet loss only sees differences of states, the last layer's bias cancels; both sides
            # report summation noise -> compare absolutely, on the scale of that layer's weight gradient
            scale = float(ref["grads"]["model.fc.1.weight"].double().norm())
        assert float((g64[off:off + p.numel()] - gref).norm()) <= 2e-3 * max(scale, 1e-30), nm
    # eval-mode states (first view) of a sample on the updated model
    srl.model.eval()
    sel = torch.tensor([0, 77, 127])
    with torch.no_grad():
        st = srl.model.getStates(obs[sel].cuda()).cpu()
    upd = T.clone_state(OrderedDict((k, v.detach().cpu()) for k, v in srl.model.state_dict().items()), requires_grad=False)
    ref_st = T.get_states(upd, obs[sel], "triplet")
    assert float((st - ref_st).abs().max()) <= 1e-4 * float(ref_st.abs().max())
    # determinism
    srl2, _, vals2, total2, grad2 = run()
    assert vals2 == vals and total2 == total and torch.equal(grad, grad2)
    for k, v in srl2.model.state_dict().items():
        assert torch.equal(v, got[k]), k


def test_pretrained_trunk_weights_are_loaded(tmp_path, monkeypatch):
    """`resnet18(pretrained=True)` (reference models/triplet.py:16): a state_dict with torchvision's resnet18 keys — as the published
    file has them: no num_batches_tracked entries, fc 1000 x 512 — given through SRLZ_RESNET18_WEIGHTS ends up in the trunk, the
    1000-way fc is replaced by the 128-unit embedding layer, the trunk stays frozen, and the trunk's output on those weights equals
    the oracle's."""
    import preprocessing.preprocess as pre
    from models.modules import SRLModules
    from models.triplet import ResNet18Trunk
    from oracle import torch_twin as T
    from srlz import hotpath
    _threads()
    g = torch.Generator().manual_seed(18)
    tv = OrderedDict()
    for k, v in ResNet18Trunk().state_dict().items():
        if k.endswith("num_batches_tracked"):
            continue  # (torchvision 0.2.1's resnet18-5c106cde.pth predates that buffer)
        if k.endswith("running_var"):
            tv[k] = torch.rand(v.shape, generator=g) + 0.5
        elif k.endswith("running_mean") or k.endswith(".bias"):
            tv[k] = torch.randn(v.shape, generator=g) * 0.1
        elif "bn" in k or "downsample.1" in k:
            tv[k] = torch.rand(v.shape, generator=g) + 0.5  # BatchNorm weight
        else:
            tv[k] = torch.randn(v.shape, generator=g) * (2.0 / (v[0].numel())) ** 0.5
    assert tuple(tv["fc.weight"].shape) == (1000, 512) and "layer2.0.downsample.0.weight" in tv and len(tv) == 102
    path = str(tmp_path / "resnet18-like.pth")
    torch.save(tv, path)
    monkeypatch.setenv("SRLZ_RESNET18_WEIGHTS", path)
    pre.N_CHANNELS = 9
    np.random.seed(2)
    torch.manual_seed(2)
    model = SRLModules(state_dim=16, action_dim=6, cuda=True, model_type="custom_cnn", losses=["triplet"])
    sd = model.state_dict()
    for k, v in tv.items():
        if k.startswith("fc."):
            continue
        assert torch.equal(sd["model.conv_layers." + k], v), k
    assert tuple(sd["model.conv_layers.fc.weight"].shape) == (128, 512)  # replaced AFTER loading (reference triplet.py:20-22)
    frozen = [n for n, p in model.named_parameters() if n.startswith("model.conv_layers.") and not n.startswith("model.conv_layers.fc")]
    assert frozen and all(not dict(model.named_parameters())[n].requires_grad for n in frozen)
    assert dict(model.named_parameters())["model.conv_layers.fc.weight"].requires_grad
    x = torch.from_numpy(gu.synthetic_obs(2, 3, 181)[0])
    ref = T.resnet18_features(T.clone_state(sd, requires_grad=False), x, False)
    model = model.to("cuda").eval()
    feat = hotpath.resnet18_forward(model.model.conv_layers, x.cuda(), False)
    assert float((feat.cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    # without the variable the seeded random initialisation stays (there is no download in this build)
    monkeypatch.delenv("SRLZ_RESNET18_WEIGHTS")
    np.random.seed(2)
    torch.manual_seed(2)
    plain = SRLModules(state_dim=16, action_dim=6, cuda=True, model_type="custom_cnn", losses=["triplet"])
    assert not torch.equal(plain.state_dict()["model.conv_layers.conv1.weight"], tv["conv1.weight"])
