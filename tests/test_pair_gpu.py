"""The batched pair (GPU): `SRLModules.forwardPair(obs, next_obs)` — ONE model call over [obs ; next_obs] with two
BatchNorm groups (include/srlz.h: `groups`) — against the two separate calls `model(obs)`, `model(next_obs)` the
reference makes (models/learner.py:392-393).

Forward: every output, every BatchNorm running statistic and counter must be BIT-IDENTICAL (per group the kernels do
exactly the arithmetic of a single-group launch; the groups' momentum updates happen in call order).  Backward: the
weight-gradient kernels sum over both groups in one launch, so parameter gradients agree up to summation order (1e-5 of
the tensor's largest entry).
"""
from collections import OrderedDict

import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


def build(losses, C=3, S=40, seed=3, split=None, inverse="linear"):
    import preprocessing.preprocess as pre
    from models.modules import SRLModules, SRLModulesSplit
    pre.N_CHANNELS = C
    np.random.seed(seed)
    torch.manual_seed(seed)
    if split is not None:
        m = SRLModulesSplit(state_dim=S, action_dim=6, cuda=True, model_type="custom_cnn", losses=losses,
                            split_dimensions=split, inverse_model_type=inverse)
    else:
        m = SRLModules(state_dim=S, action_dim=6, cuda=True, model_type="custom_cnn", losses=losses, inverse_model_type=inverse)
    return m.to("cuda")  # (SRLModules.cuda is the reference's bool attribute, not nn.Module.cuda)


def run(model, losses, obs, nxt, act, pair, eps=None):
    """forward (pair or two calls) + the loop body's losses + backward -> (outputs, buffers, grads)."""
    import losses.losses as L
    init = OrderedDict((k, v.detach().clone()) for k, v in model.state_dict().items())
    model.train()
    for p in model.parameters():
        p.grad = None
    if eps is not None:
        it = iter(eps)
        model.model.eps_fn = lambda mu: next(it).to(mu.device)
    if pair:
        a, b = model.forwardPair(obs, nxt)
    else:
        a, b = model(obs), model(nxt)
    lm = L.LossManager(model, None)
    outs = OrderedDict()
    if "vae" in losses:
        (dec, mu, logvar), (ndec, nmu, nlogvar) = a, b
        states, next_states = model.getStates(obs), model.getStates(nxt)
        L.kullbackLeiblerLoss(mu, nmu, logvar, nlogvar, loss_manager=lm, beta=1.0)
        L.generationLoss(dec, ndec, obs, nxt, weight=0.5e-6, loss_manager=lm)
        outs.update(dec=dec, ndec=ndec, mu=mu, nmu=nmu, logvar=logvar, nlogvar=nlogvar)
    elif "autoencoder" in losses:
        (states, dec), (next_states, ndec) = a, b
        L.autoEncoderLoss(obs, dec, nxt, ndec, weight=1.0, loss_manager=lm)
        outs.update(dec=dec, ndec=ndec)
    else:
        states, next_states = a, b
    outs.update(states=states, next_states=next_states)
    if "forward" in losses:
        L.forwardModelLoss(model.forwardModel(states, act), next_states, weight=1.0, loss_manager=lm)
    if "inverse" in losses:
        L.inverseModelLoss(model.inverseModel(states, next_states), act, weight=2.0, loss_manager=lm)
    total = lm.computeTotalLoss()
    total.backward()
    torch.cuda.synchronize()
    outs = OrderedDict((k, v.detach().clone()) for k, v in outs.items())
    outs["losses"] = torch.tensor(lm.lossValues() + [float(total.detach())])
    bufs = OrderedDict((k, v.detach().clone()) for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k)
    grads = OrderedDict((k, None if p.grad is None else p.grad.detach().clone()) for k, p in model.named_parameters())
    model.load_state_dict(init)
    return outs, bufs, grads


CASES = [
    ("ae", ["autoencoder"], 2, 3, None),
    ("ae_odd", ["autoencoder", "inverse", "forward"], 3, 3, None),
    ("vae", ["vae"], 2, 3, None),
    ("vae_c6", ["vae"], 3, 6, None),
    ("cnn", ["inverse", "forward"], 4, 3, None),
    ("split_ae", ["autoencoder", "inverse", "forward"], 2, 3, OrderedDict([("autoencoder", 20), ("inverse", 20), ("forward", -1)])),
    # split + VAE: forwardVAE hands out the mu MASKED to the 'vae' split, getStates must still return the full mu (reference
    # modules.py:133-134 -> models.py:131-139) or the inverse / forward heads train on zeros (round-2 advisor finding)
    ("split_vae", ["vae", "inverse", "forward"], 2, 3, OrderedDict([("vae", 20), ("inverse", 20), ("forward", -1)])),
]


@pytest.mark.parametrize("name,losses,B,C,split", CASES, ids=[c[0] for c in CASES])
def test_forward_pair_equals_two_calls(name, losses, B, C, split):
    model = build(losses, C=C, split=split)
    o, n, a = gu.golden_inputs(B, C, 6, seed=321)
    obs, nxt, act = torch.from_numpy(o).cuda(), torch.from_numpy(n).cuda(), torch.from_numpy(a).view(-1, 1).cuda()
    eps = None
    if "vae" in losses:
        torch.manual_seed(17)
        eps = [torch.randn(B, 40), torch.randn(B, 40)]
    two = run(model, losses, obs, nxt, act, pair=False, eps=eps)
    one = run(model, losses, obs, nxt, act, pair=True, eps=eps)
    for k in two[0]:
        assert torch.equal(one[0][k], two[0][k]), "forward output %s differs (max %.3e)" % (
            k, float((one[0][k].double() - two[0][k].double()).abs().max()))
    for k in two[1]:
        assert torch.equal(one[1][k], two[1][k]), "buffer %s differs" % k
    for k, g2 in two[2].items():
        g1 = one[2][k]
        assert (g1 is None) == (g2 is None), k
        if g2 is None:
            continue
        scale = max(float(g2.abs().max()), 1e-30)
        if k in gu.NOISE_BIASES:
            wk = k.replace(".bias", ".weight")
            scale = max(float(two[2][wk].abs().max()), 1e-30)  # analytically zero: compare on the weight gradient's scale
        err = float((g1.double() - g2.double()).abs().max()) / scale
        assert err <= 1e-5, "grad %s: %.3e" % (k, err)


def test_split_vae_pair_states_are_the_unmasked_mu():
    """getStates after a batched split-VAE forward: the full mu (columns of the inverse / forward splits alive), so that
    detachSplit(states, 'inverse') is not all zeros."""
    split = OrderedDict([("vae", 20), ("inverse", 20), ("forward", -1)])
    model = build(["vae", "inverse", "forward"], split=split)
    o, n, _ = gu.golden_inputs(2, 3, 6, seed=77)
    obs, nxt = torch.from_numpy(o).cuda(), torch.from_numpy(n).cuda()
    model.train()
    (dec, mu_s, _), (_, nmu_s, _) = model.forwardPair(obs, nxt)
    states, next_states = model.getStates(obs), model.getStates(nxt)
    assert float(mu_s[:, 20:].abs().max()) == 0.0 and float(states[:, 20:].abs().max()) > 0.0
    assert torch.equal(states[:, :20], mu_s[:, :20]) and torch.equal(next_states[:, :20], nmu_s[:, :20])
    assert float(model.detachSplit(states, "inverse").abs().max()) > 0.0


def test_pair_draws_the_vae_noise_like_two_calls():
    """Default noise source (torch's generator, reference models.py:161): the batched pair consumes it exactly like the two
    model calls it stands for — same seed, same reconstructions."""
    model = build(["vae"])
    o, n, _ = gu.golden_inputs(2, 3, 6, seed=78)
    obs, nxt = torch.from_numpy(o).cuda(), torch.from_numpy(n).cuda()
    init = OrderedDict((k, v.detach().clone()) for k, v in model.state_dict().items())
    model.train()
    torch.manual_seed(123)
    with torch.no_grad():
        a, b = model(obs), model(nxt)
    model.load_state_dict(init)
    torch.manual_seed(123)
    with torch.no_grad():
        pa, pb = model.forwardPair(obs, nxt)
    assert torch.equal(a[0], pa[0]) and torch.equal(b[0], pb[0])


def test_pair_over_adjacent_halves_is_zero_copy():
    """obs / next_obs that are the two halves of one buffer (the learner's feed, bench.py) are batched without a copy, and the
    pair-aware reconstruction loss finds both wholes."""
    from srlz import ops
    buf = torch.randn(4, 3, 224, 224, device="cuda")
    a, b = buf[:2], buf[2:]
    xx = ops.pair_cat(a, b)
    assert xx.data_ptr() == buf.data_ptr() and tuple(xx.shape) == (4, 3, 224, 224)
    assert ops.pair_of(a, b) is not None and ops.pair_of(b, a) is None
    c = torch.randn(2, 3, 224, 224, device="cuda")
    yy = ops.pair_cat(a, c)
    assert yy.data_ptr() != buf.data_ptr() and torch.equal(yy[:2], a) and torch.equal(yy[2:], c)


@pytest.mark.parametrize("losses", [["autoencoder", "inverse", "forward"], ["vae"]], ids=["aeif", "vae"])
def test_train_step_pair_follows_two_call_path(losses):
    """SRL4robotics.trainStep with the batched pair (default) against the two-call route (`_use_pair = False`) over a few steps incl. a validation one."""
    import models.learner as learner
    import preprocessing.preprocess as pre
    from losses.losses import LossManager
    pre.N_CHANNELS = 3
    learner.BATCH_SIZE = 3

    def trace(use_pair):
        srl = learner.SRL4robotics(24, model_type="custom_cnn", seed=5, learning_rate=1e-4, cuda=True, losses=losses, n_actions=6,
                                   log_folder="/tmp")
        srl._use_pair = use_pair
        lm = LossManager(srl.model, None)
        rows = []
        for step in range(4):
            o, n, a = gu.golden_inputs(3, 3, 6, seed=900 + step)
            if "vae" in losses:
                torch.manual_seed(40 + step)
                it = iter([torch.randn(3, 24), torch.randn(3, 24)])
                srl.model.model.eps_fn = lambda mu: next(it).to(mu.device)
            loss = srl.trainStep(torch.from_numpy(o).cuda(), torch.from_numpy(n).cuda(), torch.from_numpy(a).view(-1, 1).cuda(), lm,
                                 validation_mode=(step == 2))
            rows.append(lm.lossValues() + [float(loss.detach())])
        torch.cuda.synchronize()
        bufs = [b.detach().clone() for b in srl.model.buffers()]
        return np.array(rows), bufs

    two, bufs2 = trace(False)
    one, bufs1 = trace(True)
    assert np.array_equal(one[0], two[0])  # the first step starts from identical parameters: bit-identical losses
    np.testing.assert_allclose(one, two, rtol=2e-5)
    for x, y in zip(bufs1, bufs2):
        if x.dtype == torch.long:
            assert int(x) == int(y)
        else:
            assert float((x.double() - y.double()).abs().max()) <= 1e-3 * max(float(y.double().abs().max()), 1e-30)


@pytest.mark.parametrize("losses,C,split", [(["autoencoder", "inverse", "forward"], 3, None), (["vae"], 3, None), (["vae"], 6, None),
                                            (["dae"], 3, None),
                                            (["autoencoder", "inverse"], 3, OrderedDict([("autoencoder", 16), ("inverse", 8)]))],
                         ids=["aeif", "vae", "vae_c6", "dae", "split_ae"])
def test_recon_loss_in_the_decoder_epilogue_follows_the_unfused_step(losses, C, split):
    """SRL4robotics.trainStep with the reconstruction / generation loss taken inside the last ConvTranspose (default) against
    the un-fused route (`hotpath._FUSE_RECON = False`: decoded frames written, loss and its gradient as separate passes): the same loss values (fp64 partial
    sums in another fixed order: equal to fp32 rounding), and — the gradient handed to the decoder being bit-identical — the
    same parameters after a few steps including a validation one."""
    import models.learner as learner
    import preprocessing.preprocess as pre
    from losses.losses import LossManager
    from srlz import hotpath
    pre.N_CHANNELS = C
    learner.BATCH_SIZE = 3

    def trace(fused):
        hotpath._FUSE_RECON = fused
        try:
            srl = learner.SRL4robotics(24, model_type="custom_cnn", seed=5, learning_rate=1e-4, cuda=True, losses=losses, n_actions=6,
                                       log_folder="/tmp", multi_view=C > 3, split_dimensions=split if split is not None else -1,
                                       losses_weights_dict=None if split is None else {k: 1.0 for k in split})
            lm = LossManager(srl.model, None)
            rows = []
            for step in range(4):
                o, n, a = gu.golden_inputs(3, C, 6, seed=900 + step)
                both = torch.from_numpy(np.concatenate((o, n), 0)).cuda()  # the learner's feed: two halves of one buffer
                obs, nxt = both[:3], both[3:]
                noisy = None
                if "dae" in losses:
                    nb = torch.from_numpy(np.concatenate((gu.golden_noisy(o, 1 + step), gu.golden_noisy(n, 2 + step)), 0)).cuda()
                    noisy = (nb[:3], nb[3:])
                if "vae" in losses:
                    torch.manual_seed(40 + step)
                    it = iter([torch.randn(3, 24), torch.randn(3, 24)])
                    srl.model.model.eps_fn = lambda mu: next(it).to(mu.device)
                loss = srl.trainStep(obs, nxt, torch.from_numpy(a).view(-1, 1).cuda(), lm, validation_mode=(step == 2),
                                     noisy_obs=None if noisy is None else noisy[0], next_noisy_obs=None if noisy is None else noisy[1])
                rows.append(lm.lossValues() + [float(loss.detach())])
                names = list(lm.names)
            torch.cuda.synchronize()
            return np.array(rows), names, srl.flat_params.flat.detach().clone(), [b.detach().clone() for b in srl.model.buffers()]
        finally:
            hotpath._FUSE_RECON = True

    plain, names0, p0, b0 = trace(False)
    fused, names1, p1, b1 = trace(True)
    assert names0 == names1
    np.testing.assert_allclose(fused[0], plain[0], rtol=3e-7)  # identical parameters: only the order of the fp64 partial sums differs
    np.testing.assert_allclose(fused, plain, rtol=2e-5)
    assert float((p1 - p0).abs().max()) <= 2e-5 * float(p0.abs().max())
    for x, y in zip(b1, b0):
        if x.dtype == torch.long:
            assert int(x) == int(y)
        else:
            assert float((x.double() - y.double()).abs().max()) <= 1e-4 * max(float(y.double().abs().max()), 1e-30)


def test_recon_loss_fusion_is_what_the_default_step_runs():
    """The product path takes the fused node (the decoded frames of the step are the error tensor, flagged as such)."""
    import models.learner as learner
    import preprocessing.preprocess as pre
    from losses.losses import LossManager
    from srlz import ops
    pre.N_CHANNELS = 3
    learner.BATCH_SIZE = 2
    srl = learner.SRL4robotics(16, model_type="custom_cnn", seed=5, learning_rate=1e-4, cuda=True, losses=["autoencoder"], n_actions=6,
                               log_folder="/tmp")
    o, n, a = gu.golden_inputs(2, 3, 6, seed=5)
    both = torch.from_numpy(np.concatenate((o, n), 0)).cuda()
    seen = []
    real = ops.DecOutLossFn.apply

    def spy(*args):
        seen.append(1)
        return real(*args)
    ops.DecOutLossFn.apply = spy
    try:
        lm = LossManager(srl.model, None)
        srl.trainStep(both[:2], both[2:], torch.from_numpy(a).view(-1, 1).cuda(), lm)
    finally:
        ops.DecOutLossFn.apply = real
    assert seen == [1] and lm.names == ["reconstruction_loss"]


def test_separately_allocated_frames_take_the_unfused_route_in_the_default_environment():
    """obs and next_obs that are NOT the two halves of one buffer (a caller outside learn()'s feed): no fused reconstruction loss is
    possible (ops.pair_of finds no pair) — the step runs the batched model call on a concatenated copy, gets real decoded frames
    and the separate pair loss.  Same environment, same weights, same inputs as the adjacent case: the loss agrees to fp32
    rounding and the parameters after the step to summation order.  On the fused route the "decoded frames" never existed:
    _forwardPair hands None to its caller (a consumer of a reconstruction must fail loudly, not read dec - target)."""
    import models.learner as learner
    import preprocessing.preprocess as pre
    from losses.losses import LossManager
    from srlz import ops
    pre.N_CHANNELS = 3
    learner.BATCH_SIZE = 3
    o, n, a = gu.golden_inputs(3, 3, 6, seed=77)
    act = torch.from_numpy(a).view(-1, 1).cuda()

    def one_step(adjacent):
        srl = learner.SRL4robotics(24, model_type="custom_cnn", seed=5, learning_rate=1e-4, cuda=True,
                                   losses=["autoencoder", "inverse"], n_actions=6, log_folder="/tmp")
        lm = LossManager(srl.model, None)
        if adjacent:
            both = torch.from_numpy(np.concatenate((o, n), 0)).cuda()
            obs, nxt = both[:3], both[3:]
        else:
            obs, nxt = torch.from_numpy(o).cuda(), torch.from_numpy(n).cuda()
            assert ops.pair_of(obs, nxt) is None
        calls = {"fused": 0, "plain": 0}
        real_fused, real_plain = ops.DecOutLossFn.apply, ops.DecOutFn.apply
        ops.DecOutLossFn.apply = lambda *args: (calls.__setitem__("fused", calls["fused"] + 1), real_fused(*args))[1]
        ops.DecOutFn.apply = lambda *args: (calls.__setitem__("plain", calls["plain"] + 1), real_plain(*args))[1]
        try:
            srl.model.train()
            out = srl._forwardPair(obs, nxt, (obs, nxt, True))
            loss = srl.trainStep(obs, nxt, act, lm)
        finally:
            ops.DecOutLossFn.apply, ops.DecOutFn.apply = real_fused, real_plain
        torch.cuda.synchronize()
        return srl, out, calls, lm.lossValues() + [float(loss.detach())]

    srl_a, out_a, calls_a, loss_a = one_step(True)
    srl_s, out_s, calls_s, loss_s = one_step(False)
    assert calls_a == {"fused": 2, "plain": 0} and calls_s == {"fused": 0, "plain": 2}
    # fused route: (states, None) twice + the loss; plain route: real reconstructions, no loss
    assert out_a[2] is not None and out_a[0][1] is None and out_a[1][1] is None and out_a[0][0].shape == (3, 24)
    assert out_s[2] is None and tuple(out_s[0][1].shape) == (3, 3, 224, 224) and torch.isfinite(out_s[1][1]).all()
    np.testing.assert_allclose(loss_s, loss_a, rtol=3e-6)
    p_a, p_s = srl_a.flat_params.flat, srl_s.flat_params.flat
    assert float((p_a - p_s).abs().max()) <= 2e-5 * float(p_a.abs().max())
