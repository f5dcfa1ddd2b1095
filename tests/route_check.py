"""The check behind tests/test_default_route_gpu.py and tests/test_fullsize_gpu.py::test_full_size_gradient_bucket (GPU):
ONE `SRL4robotics.trainStep` exactly as `learn()` / `bench.py` run it (batched pair with two BatchNorm groups, every fusion on),
observed through srlz.hotpath.OBSERVE — which only keeps references to tensors the default route saves anyway — and compared, on the
gradient BUCKET Adam consumes (srlz.optim.FlatParams.grad), parameter by parameter, with the oracle's backward evaluated at the step's
own ReLU / max-pool decisions.

Reference: the loop body models/learner.py:373-497 of /root/reference (one `loss.backward()` over the whole minibatch, :489),
restated by oracle/torch_twin.py::train_step.
"""
import time
from collections import OrderedDict

import torch

import golden_util as gu

NOISE_GRADS = ("decoder_conv.0.bias", "decoder_conv.3.bias", "decoder_conv.6.bias", "decoder_conv.9.bias")


def rel(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def check_default_route_bucket(losses, B, C, rtol=1e-4, seed=1234, oracle_dtype=torch.float64, expect_launched=(), report=None):
    """Runs the step, asserts it took the default route, and holds losses and the gradient bucket to `rtol` against the
    decision-pinned oracle in `oracle_dtype`.  Returns {"worst": {param: err}, "launched": set, "oracle_s": seconds}."""
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
    init_o = OrderedDict((k, v.to(oracle_dtype).cpu() if v.is_floating_point() else v.clone().cpu()) for k, v in init.items())
    obs_np, next_obs_np, actions = gu.golden_inputs(B, C, 6, seed=seed)
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
    for k in expect_launched:
        assert k in launched, (k, sorted(launched))
    pins = gu.pins_from_observed(observed, B)
    got_losses = dict(zip(lm.names, [float(v) for v in lm.lossValues()]))
    got_total = float(total.detach())
    flat = srl.flat_params
    named = [(n, p) for n, p in srl.model.named_parameters() if p.requires_grad]
    assert len(named) == len(flat.params) and all(p is q for (_, p), q in zip(named, flat.params))
    got = OrderedDict((k, flat.grad[off:off + p.numel()].view(p.shape).detach().double().cpu())
                      for (k, p), off in zip(named, flat.offsets))
    assert srl.optimizer.steps() == 1
    del observed, srl, lm, flat, named, d_obs, d_next, total
    torch.cuda.empty_cache()

    # ---- the oracle at exactly those decisions
    t0 = time.time()
    sd_o = T.clone_state(init_o)
    ref = T.train_step(sd_o, losses, obs.to(oracle_dtype), next_obs.to(oracle_dtype), actions,
                       eps=None if eps is None else eps[0].to(oracle_dtype),
                       next_eps=None if eps is None else eps[1].to(oracle_dtype), pins=pins)
    oracle_s = time.time() - t0
    for k, v in ref["losses"].items():
        assert abs(got_losses[k] - v) <= rtol * max(abs(v), 1e-6), (k, got_losses[k], v)
    assert abs(got_total - ref["total"]) <= rtol * abs(ref["total"])

    # ---- the bucket Adam consumed, parameter by parameter
    checked, worst = 0, OrderedDict()
    for k, g in got.items():
        gref = ref["grads"].get(k)
        if gref is None:
            assert float(g.abs().max()) == 0.0, k
            continue
        if k.endswith(NOISE_GRADS):  # analytically zero (a bias in front of a train-mode BatchNorm): summation noise on both sides
            scale = ref["grads"][k.replace(".bias", ".weight")].abs().max().item()
            assert float((g - gref.double()).abs().max()) < 1e-4 * scale, k
            continue
        e = rel(g, gref)
        worst[k] = e
        assert e <= rtol, "default route, B = %d, grad %s: %.3e vs the decision-pinned %s oracle" % (B, k, e, oracle_dtype)
        checked += 1
    assert checked >= 10
    out = {"worst": worst, "launched": launched, "oracle_s": oracle_s, "losses": got_losses}
    if report is not None:
        report(out)
    return out
