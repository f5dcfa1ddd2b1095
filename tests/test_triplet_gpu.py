"""`--multi-view --losses triplet` on the GPU (SURVEY.md 8f-4 ii): the frozen ResNet-18 trunk (forward-only convN / BatchNorm /
residual kernels), the EmbeddingNet head (Linear - PReLU - Linear with backward), tripletLoss and the triplet branch of the
loop body, against the CPU oracle (oracle/torch_twin.py).

PARITY UNPINNED for the trunk: torchvision (where the reference gets ResNet-18 from, models/triplet.py:16) is neither under
/root/reference nor installed, so there are no reference fixtures — the oracle restates torchvision's published definition
(and is cross-checked against the module tree in tests/test_oracle_golden.py).  tripletLoss itself is pinned to the
reference (loss_kats.npz).  Weights are the seeded random initialisation (no pre-trained download is possible).
"""
import json
import os

import numpy as np
import pytest
import torch

import golden_util as gu
from dataset_util import make_dataset

pytestmark = pytest.mark.gpu


def _views(B, seed):
    """obs / next_obs with 9 channels: anchor, positive, negative views (three independent synthetic frames each)."""
    frames = [gu.synthetic_obs(B, 3, seed + i) for i in range(3)]
    obs = np.concatenate([f[0] for f in frames], axis=1)
    nxt = np.concatenate([f[1] for f in frames], axis=1)
    return torch.from_numpy(obs), torch.from_numpy(nxt)


def _build(S=16, seed=2, losses=("triplet",)):
    import preprocessing.preprocess as pre
    from models.modules import SRLModules
    pre.N_CHANNELS = 9
    np.random.seed(seed)
    torch.manual_seed(seed)
    return SRLModules(state_dim=S, action_dim=6, cuda=True, model_type="custom_cnn", losses=list(losses))


@pytest.mark.parametrize("training", [True, False], ids=["train_bn", "eval_bn"])
def test_trunk_and_embedding_match_oracle(training):
    from oracle import torch_twin as T
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    model = _build()
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    obs, _ = _views(3, 11)
    x = obs[:, :3].contiguous()
    sd = T.clone_state(init, requires_grad=False)
    ref = T.embedding_forward(sd, x, training)
    ref_feat = T.resnet18_features(T.clone_state(init, requires_grad=False), x, training)
    model = model.to("cuda")
    model.train(training)
    from srlz import hotpath
    feat = hotpath.resnet18_forward(model.model.conv_layers, x.cuda(), training)
    model.load_state_dict(init)  # (that call moved the running statistics: start the embedding pass from the same state)
    got = model.model(x.cuda())
    torch.cuda.synchronize()
    assert not feat.requires_grad
    err_f = float((feat.cpu() - ref_feat).abs().max() / ref_feat.abs().max())
    err = float((got.detach().cpu() - ref).abs().max() / ref.abs().max())
    assert err_f < 1e-4 and err < 1e-4, (err_f, err)
    after = model.state_dict()
    worst = 0.0
    for k in sd:
        if "running_" in k:
            r, g = sd[k].double(), after[k].double().cpu()
            e = float((g - r).abs().max() / max(float(r.abs().max()), 1e-30))
            worst = max(worst, e)
            assert e < 1e-4, (k, e)
        elif "num_batches_tracked" in k:
            assert int(after[k]) == int(sd[k]) == (1 if training and "conv_layers" in k else 0), k
    # getStates = the first view only (triplet.py:33-39)
    if not training:
        st = model.getStates(obs.cuda())
        assert float((st.detach().cpu() - ref).abs().max() / ref.abs().max()) < 1e-4


def test_triplet_loss_kat_and_gradient():
    """tripletLoss against the reference's value (loss_kats.npz) and torch autograd of the same expression."""
    import losses.losses as L
    from oracle import torch_twin as T
    g = gu.load("loss_kats")
    s, p, n = (torch.from_numpy(g["in/tri_" + k]).cuda().requires_grad_(True) for k in "spn")

    class M(torch.nn.Module):
        pass
    lm = L.LossManager(M(), None)
    out = L.tripletLoss(s, p, n, 1.0, lm, alpha=0.2)
    assert abs(float(out) - float(g["triplet_w1"])) <= 1e-6 * abs(float(g["triplet_w1"]))
    (3.0 * out).backward()
    sr, pr, nr = (torch.from_numpy(g["in/tri_" + k]).requires_grad_(True) for k in "spn")
    (3.0 * T.triplet_loss(sr, pr, nr, 0.2)).backward()
    for a, b in ((s, sr), (p, pr), (n, nr)):
        assert float((a.grad.cpu() - b.grad).abs().max()) <= 1e-6 * float(b.grad.abs().max())


def test_triplet_train_steps_ride_along_the_oracle():
    """SRL4robotics.trainStep with losses triplet + inverse + forward, step by step against the oracle re-seeded with the
    product's state: losses, the gradient bucket (head parameters; the trunk's stay exactly zero), Adam, the trunk's BatchNorm
    running statistics after SIX train-mode passes per step, and a validation step."""
    import models.learner as learner
    import preprocessing.preprocess as pre
    from losses.losses import LossManager
    from oracle import torch_twin as T
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    pre.N_CHANNELS = 9
    B, S, LR = 3, 16, 1e-3
    learner.BATCH_SIZE = B
    losses = ["triplet", "inverse", "forward"]
    srl = learner.SRL4robotics(S, model_type="custom_cnn", seed=4, learning_rate=LR, cuda=True, multi_view=True, losses=losses,
                               n_actions=6, log_folder="/tmp")
    lm = LossManager(srl.model, None)
    fp, opt = srl.flat_params, srl.optimizer
    pname = {id(p): n for n, p in srl.model.named_parameters()}
    assert all(not n.startswith("model.conv_layers.layer") for n in (pname[id(p)] for p in fp.params))  # frozen: not in the bucket
    taken = {}
    real = opt.step

    def spy(scale=1.0):
        fp.deliver()
        taken["grad"] = fp.grad.detach().clone()
        return real(scale)
    opt.step = spy
    for step in range(3):
        validation = step == 1
        obs, nxt = _views(B, 50 + 10 * step)
        act = torch.from_numpy(np.random.RandomState(step).randint(0, 6, (B,)).astype(np.int64))
        before = {k: v.detach().cpu().clone() for k, v in srl.model.state_dict().items()}
        flat0, m0, v0, t0 = fp.flat.clone(), opt.m.clone(), opt.v.clone(), opt.steps()
        taken.clear()
        loss = srl.trainStep(obs.cuda(), nxt.cuda(), act.view(-1, 1).cuda(), lm, validation_mode=validation)
        rec = dict(zip(lm.names, lm.lossValues()))
        rec["total"] = float(loss.detach())
        torch.cuda.synchronize()
        sd = T.clone_state(before)
        ref = T.train_step(sd, losses, obs, nxt, act, training=not validation)
        assert sorted(rec) == sorted(list(ref["losses"]) + ["total"])
        for nm, v in list(ref["losses"].items()) + [("total", ref["total"])]:
            assert abs(rec[nm] - v) <= 2e-5 * max(abs(v), 1e-6), (step, nm, rec[nm], v)
        after = srl.model.state_dict()
        for k in before:
            if "running_" in k:
                r, g = sd[k].double(), after[k].double().cpu()
                assert float((g - r).abs().max()) <= 1e-4 * max(float(r.abs().max()), 1e-30), (step, k)
            elif "num_batches_tracked" in k:
                assert int(after[k]) == int(sd[k]), (step, k, int(after[k]), int(sd[k]))
        if validation:
            assert "grad" not in taken and torch.equal(fp.flat, flat0) and opt.steps() == t0
            continue
        grad = taken["grad"].double().cpu()
        for p, off in zip(fp.params, fp.offsets):
            nm = pname[id(p)]
            gref = ref["grads"].get(nm)
            got = grad[off:off + p.numel()]
            if gref is None:
                assert float(got.abs().max()) == 0.0, nm
                continue
            gref = gref.double().reshape(-1)
            assert float((got - gref).norm()) <= 2e-3 * max(float(gref.norm()), 1e-30), (step, nm)
        t = t0 + 1
        g64 = taken["grad"].double()
        m1 = 0.9 * m0.double() + 0.1 * g64
        v1 = 0.999 * v0.double() + 0.001 * g64 * g64
        expect = flat0.double() - (LR / (1 - 0.9 ** t)) * m1 / ((v1 / (1 - 0.999 ** t)).sqrt() + 1e-8)
        assert float((fp.flat.double() - expect).abs().max()) <= 3e-7


@pytest.mark.parametrize("B", [2, 5])
def test_six_views_in_one_trunk_pass_are_six_trunk_calls(B):
    """Round 6: the six trunk calls of a time-contrastive step (reference models/learner.py:383-391 via modules.py:92-100) run as ONE
    batched pass with six BatchNorm groups (hotpath.resnet18_forward(groups=6): `groups` in srlz_convn_fwd / srlz_bn_finalize_chunks /
    srlz_bn_add_relu and in the auto-encoder's conv1 / pooling / conv2 kernels).  Same tiles, same accumulation order, per-group
    statistics staged with the single-call geometry: features, every running statistic after its six momentum updates in call order
    and num_batches_tracked are BIT-IDENTICAL to six separate calls — and the whole training step (losses, gradient bucket,
    parameters after Adam) is the same step."""
    import models.learner as learner
    import preprocessing.preprocess as pre
    from losses.losses import LossManager
    from srlz import hotpath
    pre.N_CHANNELS = 9
    model = _build().to("cuda")
    model.train()
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    obs, nxt = _views(B, 23)
    views = [v.contiguous().cuda() for v in (obs[:, :3], obs[:, 3:6], obs[:, 6:], nxt[:, :3], nxt[:, 3:6], nxt[:, 6:])]
    trunk = model.model.conv_layers
    sep = torch.cat([hotpath.resnet18_forward(trunk, v, True) for v in views])
    after_sep = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.load_state_dict(init)
    one = hotpath.resnet18_forward(trunk, torch.cat(views), True, groups=6)
    torch.cuda.synchronize()
    assert torch.equal(one, sep)
    after_one = model.state_dict()
    moved = 0
    for k in init:
        assert torch.equal(after_one[k], after_sep[k]), k
        if "running_mean" in k and "conv_layers" in k:
            moved += int(not torch.equal(after_one[k], init[k]))
        if "num_batches_tracked" in k and "conv_layers" in k:
            assert int(after_one[k]) == int(init[k]) + 6, k
    assert moved == 20  # every BatchNorm of the trunk (stem + 16 in the blocks + 3 downsample branches)
    # eval mode: one record for all groups
    model.load_state_dict(init)
    model.eval()
    with torch.no_grad():
        e_one = hotpath.resnet18_forward(trunk, torch.cat(views), False, groups=6)
        e_sep = torch.cat([hotpath.resnet18_forward(trunk, v, False) for v in views])
    assert torch.equal(e_one, e_sep)

    # ---- the training step: batched (default) against the six separate calls (_use_pair = False)
    learner.BATCH_SIZE = B
    act = torch.from_numpy(np.random.RandomState(B).randint(0, 6, (B,)).astype(np.int64)).view(-1, 1).cuda()
    out = []
    for use_pair in (True, False):
        srl = learner.SRL4robotics(16, model_type="custom_cnn", seed=4, learning_rate=1e-3, cuda=True, multi_view=True,
                                   losses=["triplet", "inverse"], n_actions=6, log_folder="/tmp")
        srl._use_pair = use_pair
        lm = LossManager(srl.model, None)
        loss = srl.trainStep(obs.cuda(), nxt.cuda(), act, lm)
        torch.cuda.synchronize()
        out.append((float(loss.detach()), lm.lossValues(), srl.flat_params.grad.clone(), srl.flat_params.flat.clone(),
                    {k: v.detach().clone() for k, v in srl.model.state_dict().items()}))
    (l1, v1, g1, p1, sd1), (l0, v0, g0, p0, sd0) = out
    assert l1 == l0 and v1 == v0 and torch.equal(g1, g0) and torch.equal(p1, p0)
    assert all(torch.equal(sd1[k], sd0[k]) for k in sd0)


def test_train_cli_multi_view_triplet(tmp_path):
    """`python train.py --multi-view --losses triplet ...` end to end: 9-channel loader, EmbeddingNet, checkpoint with
    torchvision's key names, learned states from the first view."""
    import subprocess
    import sys
    make_dataset(str(tmp_path), name="tiny_mv", n_episodes=3, ep_len=14, multi_view=True)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    log = str(tmp_path / "logs" / "tri")
    args = ["--no-display-plots", "--data-folder", "tiny_mv", "--epochs", "2", "--state-dim", "6", "-bs", "4", "--multi-view",
            "--losses", "triplet", "--log-folder", log]
    proc = subprocess.run([sys.executable, os.path.join(repo, "srl-zoo_amd", "train.py")] + args, cwd=str(tmp_path),
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=400)
    assert proc.returncode == 0, proc.stdout.decode("utf-8", "replace")[-3000:]
    sd = torch.load(os.path.join(log, "srl_model.pth"), map_location="cpu")
    assert tuple(sd["model.conv_layers.layer4.1.conv2.weight"].shape) == (512, 512, 3, 3)
    assert tuple(sd["model.fc.1.weight"].shape) == (6, 128)
    z = np.load(os.path.join(log, "states_rewards.npz"))
    assert z["states"].shape == (42, 6) and np.isfinite(z["states"]).all()
    hist = np.load(os.path.join(log, "loss_history.npz"))
    assert "triplet_loss" in hist.files and np.isfinite(hist["triplet_loss"]).all()
    cfg = json.load(open(os.path.join(log, "exp_config.json")))
    assert cfg["losses"] == ["triplet"] and cfg["multi-view"] is True
    # round 5: the triplet stream is resident too — epoch 1 decodes, epoch 2 gathers [view 1 ; view 2 ; negative's view 1] by index
    e1, e2 = json.load(open(os.path.join(log, "epoch_stats.json")))
    assert e1["index_minibatches"] == 0 and e2["minibatches"] >= 2 and e2["index_minibatches"] == e2["minibatches"]
