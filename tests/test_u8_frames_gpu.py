"""The loader's uint8 frames taken by the kernels themselves (include/srlz.h: srlz_conv1_fwd_u8,
srlz_conv1_bwd_weight_fused_u8, srlz_convT_out_fwd_loss_u8, srlz_normalize_lut, srlz_normalize_u8_planar).

The contract is the one of tests/test_kernels_gpu.py::test_normalize_u8_bit_exact: ((v / 255) - mean[c]) / std[c] with the
host's three fp32 roundings (/root/reference/preprocessing/utils.py:20-32) and the loader's transpose(0, 3, 2, 1)
(/root/reference/preprocessing/data_loader.py:255).  So every consumer of the bytes must produce THE SAME BITS as the same
consumer fed with the normalised float tensor — first convolution, its fused weight gradient, the fused reconstruction /
generation loss, and the whole training step (loss terms, gradient bucket, parameters after Adam).
"""
from collections import OrderedDict

import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _frames(n, c, w, h, seed):
    return torch.from_numpy(np.random.RandomState(seed).randint(0, 256, (n, c, w, h)).astype(np.uint8))


def test_table_is_the_hosts_three_roundings():
    from preprocessing.utils import preprocessInput
    from srlz import ops
    lut = ops.norm_lut(torch.device(DEV, torch.cuda.current_device())).cpu().numpy()
    v = np.arange(256, dtype=np.float32).reshape(256, 1, 1).repeat(3, axis=2)  # a 256 x 1 "image" whose RGB value is the row index
    ref = preprocessInput(v.copy())  # [256, 1, 3]
    assert lut.shape == (3, 256) and np.array_equal(lut, ref[:, 0, :].T)


@pytest.mark.parametrize("n,c", [(3, 3), (2, 6), (2, 9)])
def test_planar_frames_as_float_is_normalize_u8(n, c):
    """frames_as_float([N,C,W,H] bytes) == normalize_u8([N,H,W,C] bytes) == the host arithmetic, bit for bit."""
    from preprocessing.utils import preprocessInput
    from srlz import ops
    nhwc = np.random.RandomState(n * 10 + c).randint(0, 256, (n, 224, 200, c)).astype(np.uint8)
    planar = np.ascontiguousarray(nhwc.transpose(0, 3, 2, 1))
    ref = np.stack([np.dstack([preprocessInput(f[..., 3 * v:3 * v + 3].astype(np.float32)) for v in range(c // 3)])
                    .transpose(2, 1, 0) for f in nhwc])
    a = ops.frames_as_float(torch.from_numpy(planar).to(DEV))
    b = ops.normalize_u8(torch.from_numpy(nhwc).to(DEV))
    assert torch.equal(a, b) and np.array_equal(a.cpu().numpy(), ref)
    f = torch.randn(2, 3, 8, 8, device=DEV)
    assert ops.frames_as_float(f) is f  # float tensors pass through
    with pytest.raises(ops.C.SrlzError):
        ops.frames_as_float(torch.zeros(2, 4, 8, 8, dtype=torch.uint8, device=DEV))


def test_frames_as_float_chunks_calls_above_the_grid_limit():
    """One srlz_normalize_u8_planar launch takes n * c <= 65535 planes (grid.y); frames_as_float splits larger calls (advisor, round 3):
    22 000 tiny RGB frames = 66 000 planes, every value through the table."""
    from srlz import ops
    dev = torch.device(DEV, torch.cuda.current_device())
    frames = torch.from_numpy(np.random.RandomState(8).randint(0, 256, (22000, 3, 4, 4)).astype(np.uint8)).to(dev)
    out = ops.frames_as_float(frames)
    lut = ops.norm_lut(dev)
    ref = torch.stack([lut[c][frames[:, c].long()] for c in range(3)], dim=1)
    assert out.shape == (22000, 3, 4, 4) and torch.equal(out, ref)
    with pytest.raises(ops.C.SrlzError):  # the C entry point itself still states its limit
        ops.C.normalize_u8_planar(ops.ptr(frames), ops.ptr(lut), ops.ptr(out), 22000, 3, 16, ops.stream())


@pytest.mark.parametrize("n,c,w,h,training", [(4, 3, 224, 224, True), (2, 6, 224, 224, True), (2, 9, 224, 224, True),
                                              (3, 3, 70, 90, True), (2, 3, 64, 64, False)])
def test_first_block_on_bytes_is_the_float_path_bit_for_bit(n, c, w, h, training):
    """ops.EncInFn (conv1 -> BatchNorm -> ReLU -> MaxPool, weight gradient with the pooling backward fused in) on uint8 frames vs on
    their normalised float tensor: raw conv output, pooled output, running statistics, dW, dgamma, dbeta all identical."""
    from srlz import ops
    x8 = _frames(n, c, w, h, 5 * n + c).to(DEV)
    xf = ops.frames_as_float(x8)
    g = torch.Generator().manual_seed(c)
    w0 = (torch.randn(64, c, 7, 7, generator=g) * 0.1)
    gamma0, beta0 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    hf, wf = (w + 6 - 7) // 2 + 1, (h + 6 - 7) // 2 + 1
    hp, wp = (hf + 2 - 3) // 2 + 1, (wf + 2 - 3) // 2 + 1
    dp = torch.randn(n, hp, wp, 64, generator=g).to(DEV)

    def run(x):
        wd, gd, bd = (t.clone().to(DEV).requires_grad_(True) for t in (w0, gamma0, beta0))
        rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
        pooled, y = ops.EncInFn.apply(x, wd, gd, bd, rm, rv, training, 1, None)
        pooled.backward(dp)
        torch.cuda.synchronize()
        return y, pooled.detach(), rm, rv, wd.grad, gd.grad, bd.grad

    for a, b in zip(run(x8), run(xf)):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n,c,hf,mean", [(4, 3, 111, True), (2, 3, 111, False), (2, 6, 111, True), (2, 3, 20, True)])
def test_fused_loss_on_byte_targets_bit_for_bit(n, c, hf, mean):
    """ops.DecOutLossFn with the observations as bytes vs as the normalised float tensor: loss, error tensor and every gradient."""
    from srlz import ops
    g = torch.Generator().manual_seed(n + c + hf)
    himg = (hf - 1) * 2 + 4
    t8 = _frames(n, c, himg, himg, 77 + c).to(DEV)
    tf = ops.frames_as_float(t8)
    y0 = torch.randn(n, hf, hf, 64, generator=g)
    st0 = None
    w0 = torch.randn(64, c, 4, 4, generator=g) * 0.05
    b0 = torch.randn(c, generator=g) * 0.1
    gamma0, beta0 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1

    def run(target):
        yp = y0.clone().to(DEV).requires_grad_(True)
        wd, bd, gd, bed = (t.clone().to(DEV).requires_grad_(True) for t in (w0, b0, gamma0, beta0))
        rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
        with ops.batch_groups(2):
            loss, err = ops.DecOutLossFn.apply(yp, st0, gd, bed, rm, rv, False, wd, bd, None, target, mean)
            (3.0 * loss).backward()
        torch.cuda.synchronize()
        return loss.detach(), err, yp.grad, wd.grad, bd.grad

    for a, b in zip(run(t8), run(tf)):
        assert torch.equal(a, b)


def _learner(losses, B, channels=3, S=200, seed=3, **kw):
    import models.learner as learner
    import preprocessing.preprocess as pre
    pre.N_CHANNELS = channels
    learner.BATCH_SIZE = B
    return learner.SRL4robotics(S, model_type="custom_cnn", seed=seed, learning_rate=1e-3, cuda=True, losses=losses,
                                n_actions=6, log_folder="/tmp", **kw)


def _one_step(losses, B, frames, as_bytes, channels=3, **kw):
    from losses.losses import LossManager
    from srlz import ops
    srl = _learner(losses, B, channels, **kw)
    if "vae" in losses:
        torch.manual_seed(5)
        eps = [torch.randn(B, 200), torch.randn(B, 200)] * 2
        it = iter(eps)
        srl.model.model.eps_fn = lambda mu: next(it).to(mu.device)
    lm = LossManager(srl.model, None)
    taken = {}
    orig = srl.optimizer.step

    def spy(grad_scale=1.0):
        srl.flat_params.deliver()
        taken["grad"] = srl.flat_params.grad.clone()
        return orig(grad_scale)
    srl.optimizer.step = spy
    both = frames.to(DEV)
    if not as_bytes:
        both = ops.frames_as_float(both)
    obs, nxt = srl._toDevicePair(both[:B], both[B:])
    taken["dtype"] = obs.dtype
    act = torch.from_numpy(np.random.RandomState(1).randint(0, 6, (B,)).astype(np.int64)).view(-1, 1).to(DEV)
    rew = torch.from_numpy(np.random.RandomState(2).randint(0, 2, (B,)).astype(np.int64)).to(DEV) if "reward" in losses else None
    total = srl.trainStep(obs, nxt, act, lm, rewards_st=rew)
    torch.cuda.synchronize()
    vals = OrderedDict(zip(lm.names, lm.lossValues()))
    return float(total.detach()), vals, taken["grad"], srl.flat_params.flat.clone(), taken["dtype"], \
        OrderedDict((k, v.detach().clone()) for k, v in srl.model.state_dict().items() if "running_" in k or "tracked" in k)


@pytest.mark.parametrize("losses,B,channels", [(["autoencoder"], 8, 3), (["vae"], 4, 3), (["autoencoder", "inverse", "forward"], 6, 3),
                                               (["inverse", "forward", "reward"], 4, 3), (["autoencoder"], 2, 6)],
                         ids=["ae", "vae", "aeif", "ifr", "ae_c6"])
def test_training_step_on_bytes_is_the_float_step_bit_for_bit(losses, B, channels):
    """SRL4robotics.trainStep fed with the loader's bytes (the learn() route) vs with the normalised float tensors: every loss term,
    the gradient bucket Adam sees, the parameters after the step and the BatchNorm buffers are identical — and the byte route is
    really taken (the observations reach the model as uint8: no normalisation pass)."""
    frames = _frames(2 * B, channels, 224, 224, 31 + B)
    a = _one_step(losses, B, frames, True, channels)
    b = _one_step(losses, B, frames, False, channels)
    assert a[4] == torch.uint8 and b[4] == torch.float32
    assert a[0] == b[0] and a[1] == b[1]
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    assert list(a[5]) == list(b[5]) and all(torch.equal(a[5][k], b[5][k]) for k in a[5])


def test_steps_that_need_floats_get_floats(monkeypatch):
    """A step with a reader of the observations that has no byte form (here: the reconstruction loss taken from the decoded frames,
    `hotpath._FUSE_RECON = False`) is handed the float tensor by _toDevicePair — same numbers again."""
    from srlz import hotpath
    B = 4
    frames = _frames(2 * B, 3, 224, 224, 9)
    ref = _one_step(["autoencoder"], B, frames, False)
    monkeypatch.setattr(hotpath, "_FUSE_RECON", False)
    got = _one_step(["autoencoder"], B, frames, True)
    assert got[4] == torch.float32
    unfused = _one_step(["autoencoder"], B, frames, False)
    assert got[0] == unfused[0] and torch.equal(got[2], unfused[2])
    assert abs(got[0] - ref[0]) <= 1e-5 * abs(ref[0])


def test_device_feed_delivers_the_pair_as_one_buffer():
    """_DeviceFeed: obs / next_obs of a loader item land as the halves of one device buffer (bytes or floats), which
    _toDevicePair hands on without a copy when the step reads bytes."""
    import models.learner as learner
    srl = _learner(["autoencoder"], 4)
    frames = _frames(8, 3, 224, 224, 1)
    items = [(0, frames[:4], frames[4:], None, None), (1, frames[4:], frames[:4], None, None)]
    feed = learner._DeviceFeed(items, srl.device)
    seen = []
    for idx, obs, nxt, _, _ in feed:
        assert obs.is_cuda and obs.dtype == torch.uint8
        assert obs.untyped_storage().data_ptr() == nxt.untyped_storage().data_ptr()
        o, no = srl._toDevicePair(obs, nxt)
        assert o.data_ptr() == obs.data_ptr() and no.data_ptr() == nxt.data_ptr() and o.dtype == torch.uint8
        seen.append((idx, o.cpu(), no.cpu()))
        feed.advance()
    assert [s[0] for s in seen] == [0, 1]
    assert torch.equal(seen[0][1], frames[:4]) and torch.equal(seen[0][2], frames[4:]) and torch.equal(seen[1][1], frames[4:])


@pytest.mark.timeout(600)
def test_full_size_step_on_bytes_bit_for_bit():
    """BASELINE.json configs[1] at its full size (bs = 256: 512 frames of 224x224x3 per step) fed with bytes vs floats."""
    B = 256
    frames = _frames(2 * B, 3, 224, 224, 2024)
    a = _one_step(["autoencoder"], B, frames, True)
    b = _one_step(["autoencoder"], B, frames, False)
    assert a[4] == torch.uint8 and a[0] == b[0] and a[1] == b[1]
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    assert all(torch.equal(a[5][k], b[5][k]) for k in a[5])


@pytest.mark.timeout(600)
@pytest.mark.parametrize("flag", ["--u8-resident", "--host-input", "--host-input-nhwc"])
def test_bench_input_modes(flag):
    """bench.py's input modes (frames as bytes resident in HBM / arriving over PCIe every step / the round-2 NHWC route) run the same
    step: one JSON line, the contract's fields, a finite loss."""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    proc = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--steps", "3", "--warmup", "1", "--batch-size", "8",
                           "--no-cpu-baseline", "--no-kernel-timers", "--allow-short", flag], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=500)
    assert proc.returncode == 0, proc.stderr.decode("utf-8", "replace")[-3000:]
    lines = [l for l in proc.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["unit"] == "images/s" and out["dtype"] == "f32"
    assert np.isfinite(out["config"]["final_loss"])
    assert ("uint8" in out["data"]) and out["data"].startswith("synthetic")
    # the workload string says where the frames of a step come from (and never "resident" for the PCIe-inclusive modes)
    w = out["config"]["workload"]
    assert ("pinned host memory" in w and "resident" not in w) if flag != "--u8-resident" else "uint8 frames [B,C,W,H] resident in HBM" in w
    assert out["timed_region_s"] > 0 and "vae" not in out
