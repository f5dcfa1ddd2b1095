"""The DEFAULT route of the product under the tight gradient check (GPU).

tests/test_step_gpu.py holds the gradients to 1e-4 against the fp64 oracle evaluated at the HIP path's own ReLU / max-pool decisions —
but it reads those decisions through srlz.hotpath.TAPS, and TAPS switches the route: no PoolLink (BatchNorm-backward sums out of the
next convolution's data-gradient epilogue), no deferred decoder BatchNorm backward, hence no fused ConvTranspose block backward, and
no loss inside the last ConvTranspose.  Here the step is `SRL4robotics.trainStep` exactly as `learn()` / `bench.py` run it (batched
pair with two BatchNorm groups, every fusion on), observed through srlz.hotpath.OBSERVE, which only keeps references to tensors the
default route saves anyway: the pooled maps with their argmax bytes, the decoder blocks' raw inputs with their BatchNorm records.
The comparison is on the gradient BUCKET Adam consumes (srlz.optim.FlatParams.grad), parameter by parameter, at 1e-4 relative.

Reference: the loop body models/learner.py:373-497 of /root/reference (restated by oracle/torch_twin.py::train_step).
"""
from collections import OrderedDict

import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu
RTOL = 1e-4
NOISE_GRADS = ("decoder_conv.0.bias", "decoder_conv.3.bias", "decoder_conv.6.bias", "decoder_conv.9.bias")


def rel(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


pins_from_observed = gu.pins_from_observed


CASES = [("step_ae_b4", ["autoencoder"], 4, 3),
         ("step_vae_b4", ["vae"], 4, 3),
         ("step_aeif_b2", ["autoencoder", "inverse", "forward"], 2, 3),
         ("step_ae_c6_b2", ["autoencoder"], 2, 6)]


@pytest.mark.parametrize("name,losses,B,C", CASES)
def test_default_route_gradient_bucket_matches_decision_pinned_fp64_oracle(name, losses, B, C):
    from oracle import torch_twin as T
    import preprocessing.preprocess as pre
    from models.learner import SRL4robotics
    from losses.losses import LossManager
    from srlz import hotpath, ops
    S = 200
    pre.N_CHANNELS = C
    srl = SRL4robotics(S, model_type="custom_cnn", seed=1, learning_rate=1e-3, cuda=True, losses=list(losses), n_actions=6,
                       log_folder="/tmp", multi_view=C > 3)
    assert srl._use_pair and not srl._use_graph and hotpath.TAPS is None
    init = OrderedDict((k, v.detach().clone()) for k, v in srl.model.state_dict().items())
    init64 = OrderedDict((k, v.double().cpu() if v.is_floating_point() else v.clone().cpu()) for k, v in init.items())
    obs_np, next_obs_np, actions = gu.golden_inputs(B, C, 6, seed=1234)
    obs, next_obs, actions = torch.from_numpy(obs_np), torch.from_numpy(next_obs_np), torch.from_numpy(actions)
    eps = None
    if "vae" in losses:
        torch.manual_seed(99)
        eps = [torch.randn(B, S), torch.randn(B, S)]
        it = iter(eps)
        srl.model.model.eps_fn = lambda mu: next(it).to(mu.device)  # one draw per model call, in call order

    dev = srl.device
    d_obs, d_next = srl._toDevicePair(obs.to(dev), next_obs.to(dev))
    lm = LossManager(srl.model, None)
    ops.timers_enable(True)
    hotpath.OBSERVE = {}
    try:
        total = srl.trainStep(d_obs, d_next, actions.view(-1, 1).to(dev), lm)
        observed = hotpath.OBSERVE
    finally:
        hotpath.OBSERVE = None
    torch.cuda.synchronize()
    launched = set(k.split("/")[0] for k in ops.timers_report())
    ops.timers_enable(False)
    # ---- it WAS the default route: pooled-block sums from the next convolution's data gradient, and (with a decoder) the fused block
    # backward and the loss inside the last ConvTranspose
    assert "conv64_dgrad_poolsum_kernel" in launched, launched
    decoder = "autoencoder" in losses or "vae" in losses
    if decoder:
        assert "conv64_bwd_fused_kernel" in launched and "convT_out_os_bwd_kernel" in launched, launched
        assert len(observed["decoder_conv.12"][1]) == 4  # DecOutLossFn's node: (y_prev, bnp, w, err)
        assert ("reconstruction_loss" if "autoencoder" in losses else "generation_loss") in lm.names
    pins = pins_from_observed(observed, B)

    # ---- the fp64 oracle at exactly those decisions
    sd64 = T.clone_state(init64)
    ref = T.train_step(sd64, losses, obs.double(), next_obs.double(), actions, eps=None if eps is None else eps[0].double(),
                       next_eps=None if eps is None else eps[1].double(), pins=pins)
    got_losses = dict(zip(lm.names, [float(v) for v in lm.lossValues()]))
    for k, v in ref["losses"].items():
        assert abs(got_losses[k] - v) <= RTOL * max(abs(v), 1e-6), (k, got_losses[k], v)
    assert abs(float(total.detach()) - ref["total"]) <= RTOL * abs(ref["total"])

    # ---- the bucket Adam consumed, parameter by parameter
    flat = srl.flat_params
    named = [(n, p) for n, p in srl.model.named_parameters() if p.requires_grad]
    assert len(named) == len(flat.params) and all(p is q for (_, p), q in zip(named, flat.params))
    checked = 0
    for (k, p), off in zip(named, flat.offsets):
        g = flat.grad[off:off + p.numel()].view(p.shape)
        gref = ref["grads"].get(k)
        if gref is None:
            assert float(g.abs().max()) == 0.0, k
            continue
        if k.endswith(NOISE_GRADS):  # analytically zero (a bias in front of a train-mode BatchNorm): summation noise on both sides
            scale = ref["grads"][k.replace(".bias", ".weight")].abs().max().item()
            assert float((g.double().cpu() - gref).abs().max()) < 1e-4 * scale, k
            continue
        e = rel(g, gref)
        assert e <= RTOL, "default route, grad %s: %.3e vs the decision-pinned fp64 oracle" % (k, e)
        checked += 1
    assert checked >= 10
    assert srl.optimizer.steps() == 1
