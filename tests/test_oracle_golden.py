"""Pins the CPU oracle (oracle/torch_twin.py) and the build's parameter construction against the golden fixtures that
tools/make_golden.py captured from the UNMODIFIED reference (/root/reference, imported in the build container).

Runs anywhere (no GPU, no /root/reference needed): this is what makes the oracle trustworthy on the GPU box.
"""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import torch_twin as T

RTOL = 2e-5


def build(losses, C=3, S=200, A=6, seed=1, inverse="linear", split=None):
    import preprocessing.preprocess as pre
    from models.modules import SRLModules, SRLModulesSplit
    pre.N_CHANNELS = C
    np.random.seed(seed)
    torch.manual_seed(seed)
    if split is not None:
        return SRLModulesSplit(state_dim=S, action_dim=A, cuda=False, model_type="custom_cnn", losses=losses,
                               split_dimensions=split, inverse_model_type=inverse)
    return SRLModules(state_dim=S, action_dim=A, cuda=False, model_type="custom_cnn", losses=losses,
                      inverse_model_type=inverse)


@pytest.mark.parametrize("tag,losses,C", [("ae_c3", ["autoencoder"], 3), ("vae_c3", ["vae"], 3),
                                          ("ae_c6", ["autoencoder"], 6), ("cnn_c3", ["inverse"], 3)])
def test_seeded_construction_reproduces_reference_init(tag, losses, C):
    """Same constructors, same order => same RNG stream => the reference's initial state_dict."""
    g = gu.load("init_" + tag)
    sd = build(losses, C=C).state_dict()
    assert list(sd.keys()) == [str(n) for n in g["names"]]
    for i, (k, v) in enumerate(sd.items()):
        assert str(list(v.shape)).replace(" ", "") == str(g["shapes"][i]).replace(" ", ""), k
        assert abs(float(v.double().sum()) - g["sums"][i]) <= 1e-9 * max(1.0, g["abss"][i]), k
        assert abs(float(v.double().abs().sum()) - g["abss"][i]) <= 1e-9 * max(1.0, g["abss"][i]), k


CASES = [("step_ae_b2", ["autoencoder"], 2, 3, "linear"),
         ("step_ae_b4", ["autoencoder"], 4, 3, "linear"),
         ("step_vae_b2", ["vae"], 2, 3, "linear"),
         ("step_vae_b4", ["vae"], 4, 3, "linear"),
         ("step_aeif_b2", ["autoencoder", "inverse", "forward"], 2, 3, "linear"),
         ("step_aeif_mlp_b2", ["autoencoder", "inverse", "forward"], 2, 3, "mlp"),
         ("step_ae_c6_b2", ["autoencoder"], 2, 6, "linear"),
         ("step_vae_c6_b2", ["vae"], 2, 6, "linear"),
         ("step_cnn_if_b2", ["inverse", "forward"], 2, 3, "linear")]


def run_twin(losses, B, C, inverse, n_steps=1, lr=None, S=200, split=None, weights=None, l1_reg=0.0, l2_reg=0.0,
             dae_seed=None, val_steps=(), threads=1):
    torch.set_num_threads(threads)
    model = build(losses, C=C, S=S, inverse=inverse, split=split)
    sd = T.clone_state(model.state_dict())
    dae_sd = None
    if dae_seed is not None:
        dae_sd = T.clone_state(build(["dae"], C=C, S=S, seed=dae_seed).state_dict(), requires_grad=False)
    opt = T.TwinAdam(sd, lr) if lr is not None else None
    outs = []
    for step in range(n_steps):
        obs, next_obs, actions = gu.golden_inputs(B, C, 6, seed=1234 + step)
        eps = [None, None]
        if "vae" in losses:
            torch.manual_seed(99 + step)
            eps = [torch.randn(B, S), torch.randn(B, S)]
        noisy = (None, None)
        if "dae" in losses:
            noisy = (torch.from_numpy(gu.golden_noisy(obs, seed=1234 + step)),
                     torch.from_numpy(gu.golden_noisy(next_obs, seed=4321 + step)))
        rewards = torch.from_numpy(gu.golden_rewards(B, seed=1234 + step)[1]) if "reward" in losses else None
        out = T.train_step(sd, losses, torch.from_numpy(obs), torch.from_numpy(next_obs), torch.from_numpy(actions),
                           eps=eps[0], next_eps=eps[1], weights=weights, split=split, rewards=rewards, l1_reg=l1_reg,
                           l2_reg=l2_reg, noisy=noisy, dae_sd=dae_sd, training=step not in val_steps)
        outs.append(out)
        if opt is not None and step not in val_steps:  # a validation minibatch: backward, no update (learner.py:487-497)
            opt.step(sd)
    return sd, outs


@pytest.mark.parametrize("name,losses,B,C,inverse", CASES)
def test_twin_step_matches_reference_golden(name, losses, B, C, inverse):
    g = gu.load(name)
    sd, outs = run_twin(losses, B, C, inverse)
    check_step_against_golden(g, sd, outs[0], losses, B, C)


@pytest.mark.parametrize("name", sorted(gu.ext_cases().keys()))
def test_twin_split_reward_reg_steps_match_reference_golden(name):
    """SRLModulesSplit / reward loss / l1-l2 regularisers / DAE inputs (SURVEY.md §8f-2,3) against the reference."""
    cfg = gu.ext_defaults(gu.ext_cases()[name])
    g = gu.load(name)
    sd, outs = run_twin(cfg["losses"], cfg["B"], 3, cfg["inverse"], S=cfg["S"], split=cfg["split"], weights=cfg["weights"],
                        l1_reg=cfg["l1_reg"], l2_reg=cfg["l2_reg"], dae_seed=cfg["dae_seed"])
    check_step_against_golden(g, sd, outs[0], cfg["losses"], cfg["B"], 3)


def test_twin_split_adam_trace_matches_reference():
    cfg = gu.ext_defaults(gu.ext_cases()["step_split_dae_rfi_b4"])
    g = gu.load("trace_split_dae_rfi_b4")
    sd, outs = run_twin(cfg["losses"], cfg["B"], 3, cfg["inverse"], S=cfg["S"], split=cfg["split"], weights=cfg["weights"],
                        l1_reg=cfg["l1_reg"], l2_reg=cfg["l2_reg"], n_steps=3, lr=1e-4)
    names = [str(n) for n in g["trace/names"]]
    for step, out in enumerate(outs):
        for j, nm in enumerate(names):
            v = float(g["trace/values"][step, j])
            got = out["total"] if nm == "total" else out["losses"][nm]
            assert abs(got - v) <= 5e-4 * max(abs(v), 1e-6), (step, nm, got, v)


def test_detach_split_kats():
    """detachSplit's kept columns for six split configurations: the oracle's list-of-blocks restatement AND the
    product's column range (SRLModulesSplit.splitRange) against the reference's masks."""
    from collections import OrderedDict as OD
    from models.modules import SRLModulesSplit
    g = gu.load("detach_kats")

    class Holder(object):
        pass
    checked = 0
    for ci in range(int(g["n_configs"])):
        cfg = OD((str(k), int(d)) for k, d in zip(g["cfg%d/keys" % ci], g["cfg%d/dims" % ci]))
        S = sum(v for v in cfg.values() if v > 0)
        h = Holder()
        h.split_dimensions = cfg
        for f in [f for f in g.files if f.startswith("cfg%d/mask/" % ci)]:
            index = f.split("/")[-1]
            ref = g[f].astype(np.float32)
            got = T.detach_split(cfg, torch.ones(2, S), index)[0].numpy()
            np.testing.assert_array_equal(got, ref)
            lo, hi = SRLModulesSplit.splitRange(h, index)
            rng = np.zeros(S, dtype=np.float32)
            rng[lo:hi] = 1.0
            np.testing.assert_array_equal(rng, ref)
            checked += 1
    assert checked >= 30


def check_step_against_golden(g, sd, out, losses, B, C):
    for k in [f for f in g.files if f.startswith("loss/")]:
        nm = k[len("loss/"):]
        v = float(g[k])
        got = out["total"] if nm == "total" else out["losses"][nm]
        assert abs(got - v) <= RTOL * max(abs(v), 1e-6), (k, got, v)
    gu.check_digest(out["states"], g, "states", rtol=RTOL)
    gu.check_digest(out["next_states"], g, "next_states", rtol=RTOL)
    if out["decoded"] is not None:
        gu.check_digest(out["decoded"], g, "decoded", rtol=RTOL)
        gu.check_digest(out["next_decoded"], g, "next_decoded", rtol=RTOL)
    if out["logvar"] is not None:
        gu.check_digest(out["logvar"], g, "logvar", rtol=RTOL)
    for k, grad in out["grads"].items():
        if ("grad/" + k + "/none") in g.files:
            assert grad is None, k
        elif k.endswith(("decoder_conv.0.bias", "decoder_conv.3.bias", "decoder_conv.6.bias", "decoder_conv.9.bias")):
            continue  # analytically zero gradients (bias feeding a train-mode BN): pure summation noise
        else:
            gu.check_digest(grad, g, "grad/" + k, rtol=1e-4)
    for k in [f for f in g.files if f.startswith("bn/")]:
        ref = np.asarray(g[k], dtype=np.float64)
        got = sd[k[len("bn/"):]].double().numpy()
        np.testing.assert_allclose(got, ref, rtol=RTOL, atol=1e-7)
    kind = "ae" if ("autoencoder" in losses or "dae" in losses) else ("vae" if "vae" in losses else "cnn")
    obs, _, _ = gu.golden_inputs(B, C, 6, seed=1234)
    st = T.get_states(sd, torch.from_numpy(obs), kind)
    np.testing.assert_allclose(st.double().numpy(), g["eval_states/full"], rtol=1e-4, atol=1e-5)


TRACES = [("trace_ae_b2", ["autoencoder"], {}), ("trace_vae_b2", ["vae"], {}),
          ("trace_aeif_b2", ["autoencoder", "inverse", "forward"], {}),
          ("trace10_ae_b2", ["autoencoder"], {}), ("trace10_vae_b2", ["vae"], {}),
          ("trace_val_aeif_b2", ["autoencoder", "inverse", "forward"], dict(val_steps=(1,))),
          ("trace_val_vae_b2", ["vae"], dict(val_steps=(2,))),
          ("trace_ae_l1l2_b2", ["autoencoder"], dict(l1_reg=1e-5, l2_reg=1e-4)),
          # the reference's default minibatch (bs = 32, BASELINE.json configs[0]); 8 threads: a step is 64 images on the CPU
          ("trace10_ae_b32", ["autoencoder"], dict(B=32, threads=8))]


@pytest.mark.parametrize("name,losses,extra", TRACES)
def test_twin_adam_trace_matches_reference(name, losses, extra):
    """3 / 4 / 10 optimisation steps (Adam, lr 1e-4), with validation minibatches in the middle for the trace_val_*
    fixtures: per-step losses follow the reference."""
    g = gu.load(name)
    n_steps = int(g["trace/values"].shape[0])
    extra = dict(extra)
    sd, outs = run_twin(losses, extra.pop("B", 2), 3, "linear", n_steps=n_steps, lr=1e-4, **extra)
    names = [str(n) for n in g["trace/names"]]
    for step, out in enumerate(outs):
        for j, nm in enumerate(names):
            v = float(g["trace/values"][step, j])
            got = out["total"] if nm == "total" else out["losses"][nm]
            assert abs(got - v) <= 5e-4 * max(abs(v), 1e-6), (step, nm, got, v)


def test_loss_kats():
    """reconstruction / generation / KL / forward / inverse on small tensors (reference losses.py free functions)."""
    g = gu.load("loss_kats")
    Tn = lambda k: torch.from_numpy(g["in/" + k])
    a, b, c, d = Tn("a"), Tn("b"), Tn("c"), Tn("d")
    assert abs(T.reconstruction_loss(a, b).item() - float(g["reconstruction"])) < 1e-6
    ae = T.reconstruction_loss(a, b) + T.reconstruction_loss(c, d)
    assert abs(ae.item() - float(g["autoencoder_w1"])) < 1e-6
    kl = 2.0 * (T.kl_loss(Tn("mu"), Tn("lv")) + T.kl_loss(Tn("nmu"), Tn("nlv")))
    assert abs(kl.item() - float(g["kl_beta2"])) < 1e-4
    fw = T.reconstruction_loss(Tn("mu"), Tn("nmu"))
    assert abs(fw.item() - float(g["forward_w1"])) < 1e-6
    inv = 2.0 * torch.nn.functional.cross_entropy(Tn("logits"), Tn("act").view(-1))
    assert abs(inv.item() - float(g["inverse_w2"])) < 1e-6
    tri = T.triplet_loss(Tn("tri_s"), Tn("tri_p"), Tn("tri_n"), 0.2)  # the reference's tripletLoss (losses.py:360-376)
    assert abs(tri.item() - float(g["triplet_w1"])) < 1e-6


def test_embedding_net_construction_and_trunk_restatement():
    """`--losses triplet`: SRLModules swaps in EmbeddingNet (reference modules.py:71-73) — a frozen ResNet-18 with
    torchvision's state_dict keys + a trainable head.  The oracle's functional trunk (PARITY UNPINNED, see torch_twin) must at
    least agree with executing the module tree the way torchvision's ResNet.forward does, in both BatchNorm modes."""
    import torch.nn.functional as F
    torch.set_num_threads(4)
    model = build(["triplet", "inverse"], S=8)
    net = model.model
    sd = model.state_dict()
    assert "model.conv_layers.layer2.0.downsample.0.weight" in sd and "model.conv_layers.layer4.1.bn2.running_var" in sd
    assert tuple(sd["model.conv_layers.fc.weight"].shape) == (128, 512) and tuple(sd["model.fc.1.weight"].shape) == (8, 128)
    assert sd["model.fc.0.weight"].numel() == 1 and float(sd["model.fc.0.weight"]) == 0.25
    trainable = [k for k, p in model.named_parameters() if p.requires_grad and k.startswith("model.")]
    assert trainable == ["model.conv_layers.fc.weight", "model.conv_layers.fc.bias", "model.fc.0.weight", "model.fc.1.weight",
                         "model.fc.1.bias"]
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(3))

    def torchvision_forward(t, z):  # torchvision.models.resnet.ResNet.forward / BasicBlock.forward, on the real modules
        z = t.maxpool(t.relu(t.bn1(t.conv1(z))))
        for layer in (t.layer1, t.layer2, t.layer3, t.layer4):
            for b in layer:
                out = b.relu(b.bn1(b.conv1(z)))
                out = b.bn2(b.conv2(out))
                z = b.relu(out + (z if b.downsample is None else b.downsample(z)))
        return t.avgpool(z).view(z.size(0), -1)

    for training in (True, False):
        state = T.clone_state(sd, requires_grad=False)
        net.conv_layers.train(training)
        with torch.no_grad():
            ref = torchvision_forward(net.conv_layers, x)
        got = T.resnet18_features(state, x, training)
        assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
        if training:  # the twin moved its copy of the running statistics exactly like the modules moved theirs
            after = model.state_dict()
            for k in ("model.conv_layers.bn1.running_mean", "model.conv_layers.layer3.0.downsample.1.running_var"):
                assert float((state[k] - after[k]).abs().max()) <= 1e-6 * max(float(after[k].abs().max()), 1e-30), k
            assert int(state["model.conv_layers.layer4.1.bn2.num_batches_tracked"]) == 1


def test_head_kats():
    g = gu.load("head_kats")
    s, ns, act = torch.from_numpy(g["in/s"]), torch.from_numpy(g["in/ns"]), torch.from_numpy(g["in/act"])
    for inv in ("linear", "mlp"):
        sd = T.clone_state(build(["autoencoder", "inverse", "forward"], inverse=inv).state_dict(), requires_grad=False)
        np.testing.assert_allclose(T.forward_model(sd, s, act, 6).numpy(), g[inv + "/forward"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(T.inverse_model(sd, s, ns).numpy(), g[inv + "/inverse"], rtol=1e-5, atol=1e-6)


def test_layer_trace():
    """Per-layer forward digests of the AE (conv outputs / pooled maps), B=2."""
    g = gu.load("layers_ae_b2")
    torch.set_num_threads(1)
    sd = T.clone_state(build(["autoencoder"]).state_dict(), requires_grad=False)
    obs, _, _ = gu.golden_inputs(2, 3, 6, seed=1234)
    taps = {}
    T.ae_forward(sd, torch.from_numpy(obs), True, taps=taps)
    checked = 0
    for k, v in taps.items():
        key = k[len("model."):]
        if (key + "/sum") in g.files:
            gu.check_digest(v, g, key, rtol=RTOL)
            checked += 1
    assert checked >= 10


def test_product_has_no_cpu_path():
    model = build(["autoencoder"])
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 3, 224, 224))
    from models.learner import SRL4robotics
    with pytest.raises(RuntimeError):
        SRL4robotics(10, model_type="custom_cnn", losses=["autoencoder"], cuda=False)


def test_pinned_decisions_reproduce_unpinned_gradient():
    """The oracle's decision-pinned mode (used by the GPU gradient parity test) fed with the oracle's OWN decisions
    gives the oracle's own gradient."""
    import torch.nn.functional as F
    torch.set_num_threads(4)
    model = build(["autoencoder"])
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    obs, next_obs, actions = gu.golden_inputs(2, 3, 6, seed=1234)
    obs, next_obs, actions = torch.from_numpy(obs), torch.from_numpy(next_obs), torch.from_numpy(actions)
    base = T.train_step(T.clone_state(init), ["autoencoder"], obs, next_obs, actions)

    def decisions(x):
        taps = {}
        sd = T.clone_state(init, requires_grad=False)
        T.ae_forward(sd, x, True, taps=taps)
        pins = {}
        for conv, bn, pool, pad in ((0, 1, 3, 1), (4, 5, 7, 0), (8, 9, 11, 0)):
            y = taps["model.encoder_conv.%d" % conv]
            z = F.batch_norm(y, None, None, sd["model.encoder_conv.%d.weight" % bn], sd["model.encoder_conv.%d.bias" % bn],
                             True, 0.1, 1e-5)
            pz, idx = F.max_pool2d(F.relu(z), 3, 2, pad, return_indices=True)
            pins["encoder_conv.%d" % pool] = (idx, pz > 0)
        for i in (2, 5, 8, 11):
            pins["decoder_conv.%d" % i] = taps["model.decoder_conv.%d" % i] > 0
        return pins
    pinned = T.train_step(T.clone_state(init), ["autoencoder"], obs, next_obs, actions,
                          pins=(decisions(obs), decisions(next_obs)))
    assert abs(pinned["total"] - base["total"]) < 1e-6 * abs(base["total"])
    for k, g in base["grads"].items():
        if g is None or k.endswith(("decoder_conv.0.bias", "decoder_conv.3.bias", "decoder_conv.6.bias", "decoder_conv.9.bias")):
            continue
        e = (pinned["grads"][k] - g).abs().max().item() / g.abs().max().item()
        assert e < 1e-5, (k, e)
