"""Host-side logic that mirrors the reference's plugin / CLI surface (no GPU, no kernels)."""
import argparse
import json
import os
from collections import OrderedDict, defaultdict

import numpy as np
import pytest
import torch

import golden_util as gu


def test_loss_manager_history_and_total_follow_reference():
    """LossManager semantics on the reference's KAT (tests/golden/loss_kats.npz): names, weights, weighted total and the
    epoch-history accumulation (weight * value added into the last slot)."""
    from losses.losses import LossManager
    g = gu.load("loss_kats")

    class M(torch.nn.Module):
        def __init__(self):
            super(M, self).__init__()
            self.lin = torch.nn.Linear(2, 2)
    hist = defaultdict(list)
    lm = LossManager(M(), hist)
    assert [tuple(p.shape) for p in lm.reg_params] == [(2, 2)]  # biases are excluded
    names = [str(n) for n in g["names"]]
    weights = g["weights"]
    # unweighted loss values recovered from the reference's weighted history of ONE update (history holds 2 updates)
    values = []
    hn = [str(n) for n in g["history/names"]]
    for n, w in zip(names, weights):
        values.append(float(g["history/values"][hn.index(n)]) / 2.0 / w)
    for n, w, v in zip(names, weights, values):
        lm.addToLosses(n, float(w), torch.tensor(v, dtype=torch.float32))
    total = lm.computeTotalLoss()
    assert abs(float(total) - float(g["total"])) < 1e-4 * abs(float(g["total"]))
    lm.updateLossHistory()
    lm.updateLossHistory()
    for n in names:
        assert abs(hist[n][-1] - float(g["history/values"][hn.index(n)])) < 1e-4 * abs(float(g["history/values"][hn.index(n)]))
    lm.resetLosses()
    assert lm.names == [] and lm.losses == []


def test_loss_argument_language():
    from utils import parseLossArguments
    kw = parseLossArguments(choices=["autoencoder", "inverse", "vae"], help="h")
    t = kw["type"]
    assert t("inverse") == "inverse"
    assert t("autoencoder:1:10") == ("autoencoder", 1.0, 10)
    assert t("vae:0.5") == ("vae", 0.5, 0)
    with pytest.raises(argparse.ArgumentTypeError):
        t("bogus")
    with pytest.raises(argparse.ArgumentTypeError):
        t("inverse:x:1")
    assert kw["help"].startswith("{autoencoder, inverse, vae}")


def test_train_cli_parsing_and_loss_resolution():
    import train
    parser = train.buildParser()
    a = parser.parse_args(["--data-folder", "data/foo/", "--losses", "autoencoder", "inverse", "-bs", "8", "--no-cuda"])
    assert a.batch_size == 8 and a.state_dim == 2 and a.learning_rate == 0.005 and a.epochs == 30 and a.val_size == 0.2
    losses, w, split = train.resolveLosses(a.losses, False)
    assert sorted(losses) == ["autoencoder", "inverse"] and w is None and split == -1
    b = parser.parse_args(["--data-folder", "x", "--losses", "autoencoder:1:20", "inverse:5:10"])
    losses, w, split = train.resolveLosses(b.losses, False)
    assert losses == ["autoencoder", "inverse"] and w == OrderedDict([("autoencoder", 1.0), ("inverse", 5.0)])
    assert split == OrderedDict([("autoencoder", 20), ("inverse", 10)])
    with pytest.raises(ValueError):
        train.resolveLosses(["autoencoder", ("inverse", 1.0, 0)], False)


def test_build_config_and_helpers(tmp_path):
    import train
    from utils import buildConfig, parseDataFolder
    from pipeline import getLogFolderName, saveConfig, NAN_ERROR, NO_PAIRS_ERROR
    assert (NO_PAIRS_ERROR, NAN_ERROR) == (10, 11)
    assert parseDataFolder("data/kuka_gym_test/") == "kuka_gym_test"
    a = train.buildParser().parse_args(["--data-folder", "d", "--losses", "vae", "--state-dim", "200"])
    a.data_folder = parseDataFolder(a.data_folder)
    a.losses, _, a.split_dimensions = train.resolveLosses(a.losses, False)
    cfg = buildConfig(a)
    assert list(cfg.keys()) == ["batch-size", "beta", "data-folder", "epochs", "learning-rate", "training-set-size", "log-folder",
                                "model-type", "seed", "state-dim", "knn-samples", "knn-seed", "l1-reg", "l2-reg", "losses",
                                "n-neighbors", "n-to-plot", "split-dimensions", "inverse-model-type"]
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        folder, name = getLogFolderName(cfg)
        assert folder.startswith("logs/d/") and name.endswith("_custom_cnn_ST_DIM200_vae") and os.path.isdir(folder)
        cfg["log-folder"] = folder
        saveConfig(cfg)
        assert json.load(open(folder + "/exp_config.json"))["state-dim"] == 200
    finally:
        os.chdir(cwd)


def test_save_states_files(tmp_path):
    from models.learner import BaseLearner
    states = np.arange(6, dtype=np.float32).reshape(3, 2)
    BaseLearner.saveStates(states, ["a", "b", "c"], np.array([0, 1, 0]), str(tmp_path))
    z = np.load(str(tmp_path / "states_rewards.npz"))
    assert np.array_equal(z["states"], states) and list(z["rewards"]) == [0, 1, 0]
    m = json.load(open(str(tmp_path / "image_to_state.json")))
    assert m["b"] == ["2.0", "3.0"]


def test_unsupported_configurations_are_rejected_loudly():
    from models.modules import SRLModules
    from models.learner import SRL4robotics
    with pytest.raises(NotImplementedError):
        SRLModules(state_dim=3, model_type="resnet", losses=["inverse"])
    with pytest.raises(NotImplementedError):
        SRL4robotics(3, model_type="custom_cnn", losses=["priors"], cuda=True)
    with pytest.raises(NotImplementedError):  # the reference itself crashes on this combination (SURVEY.md 8a note)
        SRL4robotics(3, model_type="custom_cnn", losses=["vae", "triplet"], cuda=True, multi_view=True)


def test_state_dict_keys_and_flat_params():
    """Parameters re-homed into one flat buffer stay nn.Parameters with the reference's keys; grads are views."""
    import preprocessing.preprocess as pre
    from models.modules import SRLModules
    from srlz import optim
    pre.N_CHANNELS = 3
    m = SRLModules(state_dim=10, action_dim=4, model_type="custom_cnn", losses=["vae"])
    keys = list(m.state_dict().keys())
    assert keys[:2] == ["forward_net.weight", "forward_net.bias"] and "model.encoder_fc2.weight" in keys
    before = {k: v.clone() for k, v in m.state_dict().items()}
    flat = optim.FlatParams(m)
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    p = dict(m.named_parameters())["model.encoder_fc1.bias"]
    assert p.data_ptr() >= flat.flat.data_ptr() and p.grad is not None
    p.grad.add_(1.0)
    assert float(flat.grad.sum()) == p.numel()
    flat.zero_grad()
    idx = [i for i, q in enumerate(flat.params) if q is p][0]
    assert float(flat.grad.abs().sum()) == 0.0 and p.grad.data_ptr() == flat.grad.data_ptr() + 4 * flat.offsets[idx]


def test_flat_params_staging_bookkeeping():
    """grad_buffer(): k-th request of a pass -> a view of stage k at the parameter's offset (fixed addresses), None when
    all stages are used; zero_grad() starts a new pass.  (deliver() itself is a HIP kernel: tests/test_step_gpu.py.)"""
    import preprocessing.preprocess as pre
    from models.modules import SRLModules
    from srlz import optim, ops
    pre.N_CHANNELS = 3
    m = SRLModules(state_dim=10, action_dim=4, model_type="custom_cnn", losses=["autoencoder"])
    flat = optim.FlatParams(m)
    p = dict(m.named_parameters())["model.encoder_conv.4.weight"]
    home, idx = p._srlz_flat
    assert home is flat and flat.params[idx] is p
    off = flat.offsets[idx]
    views = [flat.grad_buffer(idx) for _ in range(flat.NSTAGE + 1)]
    assert views[-1] is None and flat._dirty
    for k, v in enumerate(views[:-1]):
        assert v.shape == p.shape and v.data_ptr() == flat.stage.data_ptr() + 4 * (k * flat.stage.shape[1] + off)
    # the ops-side helpers: a staged buffer is recognised and withheld from autograd; anything else is passed through
    flat.zero_grad()
    assert not flat._dirty and flat._served[idx] == 0
    buf = ops._gbuf(p)
    assert ops._give(p, buf) is None and buf.data_ptr() == flat.stage.data_ptr() + 4 * off
    other = torch.zeros_like(p)
    assert ops._give(p, other) is other and ops._give(None, None) is None
    q = torch.nn.Parameter(torch.zeros(3))  # not part of a bucket: plain allocation, returned to autograd
    assert ops._give(q, ops._gbuf(q)) is not None


def test_split_model_host_logic():
    """SRLModulesSplit construction checks (reference models/modules.py:120-129) and the kept column ranges."""
    from collections import OrderedDict
    import preprocessing.preprocess as pre
    from models.modules import SRLModulesSplit
    pre.N_CHANNELS = 3
    split = OrderedDict([("autoencoder", 6), ("reward", -1), ("forward", 3), ("inverse", 1)])
    m = SRLModulesSplit(state_dim=10, action_dim=4, model_type="custom_cnn", losses=list(split.keys()), split_dimensions=split)
    assert [m.splitRange(k) for k in ("autoencoder", "reward", "forward", "inverse", "vae")] == \
        [(0, 6), (0, 6), (6, 9), (9, 10), (0, 0)]
    with pytest.raises(AssertionError):
        SRLModulesSplit(state_dim=11, action_dim=4, losses=list(split.keys()), split_dimensions=split)
    with pytest.raises(AssertionError):
        SRLModulesSplit(state_dim=10, action_dim=4, losses=["autoencoder"], split_dimensions=split)
    with pytest.raises(ValueError):
        SRLModulesSplit(state_dim=10, action_dim=4, model_type="resnet", losses=list(split.keys()), split_dimensions=split)
    bad = OrderedDict([("reward", -1), ("autoencoder", 10)])
    m2 = SRLModulesSplit(state_dim=10, action_dim=4, losses=list(bad.keys()), split_dimensions=bad)
    with pytest.raises(ValueError):
        m2.splitRange("reward")


def test_bench_line_contract_on_committed_profile():
    """The newest committed bench line (profiles/*_bench_ae_bs256.json, written by bench.py on an MI355X) carries every
    field of the driver's contract, and its roofline numbers are self-consistent."""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(f for f in glob.glob(os.path.join(root, "profiles", "*_bench_ae_bs256.json")))
    assert files
    line = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["dtype"] == "f32" and line["data"] == "synthetic" and "workload" in line["config"]
    images = 2 * line["config"]["global_batch"] * line["steps"]
    assert abs(line["value"] - images / (line["ms_per_step"] * 1e-3 * line["steps"])) < 1e-3 * line["value"]
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["algorithmic_gflop_per_launch"] / r["avg_launch_us"] * 1e3) < 0.01 * r["achieved"]
    c = line["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    if os.path.basename(files[-1]) >= "r05":  # round 5: both legs of the headline metric, and the length of the timed region
        assert line["timed_region_s"] >= 0.2 and abs(line["timed_region_s"] - line["ms_per_step"] * 1e-3 * line["steps"]) < 1e-3
        v = line["vae"]
        assert v["ms_per_step"] > 0 and v["steps"] == line["steps"] and "--losses vae" in v["workload"]
        assert abs(v["images_per_s"] - 2 * line["config"]["global_batch"] / (v["ms_per_step"] * 1e-3)) < 1e-3 * v["images_per_s"]
        assert "resident in HBM" in line["config"]["workload"]


def test_device_feed_protocol(monkeypatch):
    """_DeviceFeed (copy-stream look-ahead of the learner) yields exactly one epoch of the loader, in order, and never pulls
    from the loader again after its end-of-epoch marker — the loader process is already producing the next epoch."""
    import models.learner as learner

    class FakeStream(object):
        def __init__(self, device=None):
            pass

        def wait_event(self, ev):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    class FakeEvent(object):
        def record(self, stream):
            pass

    monkeypatch.setattr(learner.th.cuda, "Stream", FakeStream)
    monkeypatch.setattr(learner.th.cuda, "Event", FakeEvent)
    monkeypatch.setattr(learner.th.cuda, "stream", lambda s: s)
    monkeypatch.setattr(learner.th.cuda, "current_stream", lambda device=None: FakeStream())
    monkeypatch.setattr(torch.Tensor, "record_stream", lambda self, s: None, raising=False)

    class EpochLoader(object):
        """Like preprocessing.data_loader.DataLoader(infinite_loop=True): StopIteration once per epoch, then goes on."""

        def __init__(self, per_epoch):
            self.per_epoch, self.count, self.pulls = per_epoch, 0, 0

        def __iter__(self):
            return self

        def __next__(self):
            self.pulls += 1
            if self.count == self.per_epoch:
                self.count = 0
                raise StopIteration
            self.count += 1
            return (self.count, torch.full((2,), float(self.count)), None)

    loader = EpochLoader(3)
    for epoch in range(2):
        feed = learner._DeviceFeed(loader, "cpu")
        seen = []
        for idx, t, none in feed:
            seen.append((idx, float(t[0])))
            feed.advance()
            feed.advance()  # idempotent
        assert seen == [(1, 1.0), (2, 2.0), (3, 3.0)]
        assert feed.exhausted
    assert loader.pulls == 2 * 4  # three minibatches + one end marker per epoch, nothing beyond


def _fake_cuda(monkeypatch, learner):
    class FakeStream(object):
        def __init__(self, device=None):
            pass

        def wait_event(self, ev):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    class FakeEvent(object):
        def record(self, stream):
            pass

    monkeypatch.setattr(learner.th.cuda, "Stream", FakeStream)
    monkeypatch.setattr(learner.th.cuda, "Event", FakeEvent)
    monkeypatch.setattr(learner.th.cuda, "stream", lambda s: s)
    monkeypatch.setattr(learner.th.cuda, "current_stream", lambda device=None: FakeStream())
    monkeypatch.setattr(torch.Tensor, "record_stream", lambda self, s: None, raising=False)


def test_device_feed_hands_the_two_frames_over_as_one_buffer(monkeypatch):
    """(idx, obs, next_obs, ...) items: the two frames land as the halves of ONE buffer (what the batched model call and the
    pair-aware losses want), whatever their dtype; everything else in the item is moved as it is."""
    import models.learner as learner
    _fake_cuda(monkeypatch, learner)
    for dtype in (torch.uint8, torch.float32):
        obs = (torch.arange(2 * 3 * 4 * 5) % 251).reshape(2, 3, 4, 5).to(dtype)
        nxt = obs.flip(0).contiguous()
        items = [(7, obs, nxt, None, None), (8, obs, nxt[:1], None, None)]
        feed = learner._DeviceFeed(items, "cpu")
        out = []
        for item in feed:
            out.append(item)
            feed.advance()
        (i0, a, b, n0, n1), (i1, c, d, _, _) = out
        assert (i0, i1) == (7, 8) and n0 is None and n1 is None
        assert torch.equal(a, obs) and torch.equal(b, nxt) and a.dtype == dtype
        assert a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and b.storage_offset() == a.numel()
        assert torch.equal(c, obs) and torch.equal(d, nxt[:1])  # different shapes: moved separately
        assert c.untyped_storage().data_ptr() != d.untyped_storage().data_ptr()


def test_which_steps_read_the_loaders_bytes(monkeypatch):
    """SRL4robotics._readsBytes: only steps whose sole readers of the observations are conv1 and the fused reconstruction loss keep
    the uint8 frames; every configuration with another reader gets the float tensor."""
    import models.learner as learner
    from srlz import hotpath

    class Stub(object):
        model_type, _use_pair, _use_graph = "custom_cnn", True, False
        use_triplets = use_dae = use_vae = perceptual_similarity_loss = False

    reads = learner.SRL4robotics._readsBytes
    assert reads(Stub()) is True
    for attr, value in (("model_type", "resnet"), ("_use_pair", False), ("_use_graph", True),
                        ("use_triplets", True), ("use_dae", True)):
        s = Stub()
        setattr(s, attr, value)
        assert not reads(s), attr
    s = Stub()
    s.use_vae = True
    assert reads(s)
    s.perceptual_similarity_loss = True
    assert not reads(s)
    monkeypatch.setattr(hotpath, "_FUSE_RECON", False)
    assert not reads(Stub())
    monkeypatch.setattr(hotpath, "_FUSE_RECON", True)
    monkeypatch.setattr(learner, "RAW_UINT8_INPUT", False)
    assert not reads(Stub())
    # the layout of uint8 frames is STATED (frame_layout), never guessed from a shape: a [2, 3, 6, 9] byte tensor fits both readings
    class Lay(learner.BaseLearner):
        def __init__(self, layout):
            self.frame_layout = layout
    ambiguous = torch.zeros(2, 3, 6, 9, dtype=torch.uint8)
    assert Lay("planar")._isPlanar(ambiguous) and not Lay("nhwc")._isPlanar(ambiguous)
    assert Lay("planar")._isPlanar(torch.zeros(2, 9, 224, 224, dtype=torch.uint8))
    assert not Lay("planar")._isPlanar(torch.zeros(2, 3, 224, 224))  # float observations are never "bytes"
    with pytest.raises(ValueError):
        Lay("chw")._isPlanar(ambiguous)
    assert learner.BaseLearner.frame_layout == "planar"
