"""End-to-end SRL4robotics.learn() on a tiny generated dataset (GPU): loader process, train/validation split,
best-model checkpoint in the reference's format, learned-states output — and parity of those learned states with the
CPU oracle evaluated on the saved checkpoint."""
import json
import os

import numpy as np
import pytest
import torch

from dataset_util import make_dataset

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    root = tmp_path_factory.mktemp("learn")
    info = make_dataset(str(root), n_episodes=4, ep_len=26)
    cwd = os.getcwd()
    os.chdir(str(root))
    os.makedirs("logs/run", exist_ok=True)
    yield info
    os.chdir(cwd)


@pytest.mark.parametrize("losses,kind", [(["autoencoder", "inverse", "forward"], "ae"), (["vae"], "vae")])
def test_learn_end_to_end(workdir, losses, kind):
    import models.learner as learner
    import preprocessing.preprocess as pre
    from models.learner import SRL4robotics
    from oracle import torch_twin as T
    name, paths, actions, rewards, starts = workdir
    pre.N_CHANNELS = 3
    learner.N_EPOCHS, learner.BATCH_SIZE, learner.VALIDATION_SIZE, learner.DISPLAY_PLOTS = 3, 8, 0.2, False
    log = "logs/run_" + kind
    os.makedirs(log, exist_ok=True)
    srl = SRL4robotics(12, model_type="custom_cnn", seed=3, learning_rate=1e-3, cuda=True, losses=losses, n_actions=6,
                       log_folder=log)
    init_keys = list(srl.model.state_dict().keys())
    loss_history, states, pairs = srl.learn(paths, actions, rewards, starts)
    assert states.shape == (len(paths), 12) and np.isfinite(states).all()
    assert set(n for n, _ in pairs) <= {"forward_loss", "inverse_loss", "reconstruction_loss", "kl_loss", "generation_loss"}
    for k in ("train_loss", "val_loss"):
        assert len(loss_history[k]) == 3 and np.isfinite(loss_history[k]).all()
    assert loss_history["train_loss"][-1] < loss_history["train_loss"][0]  # it learns
    # checkpoint: reference format (same keys, NCHW shapes), loadable on CPU
    sd = torch.load(log + "/srl_model.pth", map_location="cpu")
    assert list(sd.keys()) == init_keys
    assert tuple(sd["model.encoder_conv.0.weight"].shape) == (64, 3, 7, 7)
    assert tuple(sd["model.decoder_conv.12.weight"].shape) == (64, 3, 4, 4)
    # learned states == oracle's eval-mode states on the saved checkpoint (first 8 frames)
    from preprocessing.data_loader import DataLoader
    obs = torch.cat([DataLoader._makeBatchElement(p) for p in paths[:8]], 0)
    ref = T.get_states(T.clone_state(sd, requires_grad=False), obs, kind)
    err = np.abs(states[:8] - ref.numpy()).max() / np.abs(ref.numpy()).max()
    assert err < 1e-4, err
    # loadSavedModel round trip through exp_config.json
    with open(log + "/exp_config.json", "w") as f:
        json.dump({"state-dim": 12, "losses": losses, "n_actions": 6, "model-type": "custom_cnn"}, f)
    srl2, cfg = SRL4robotics.loadSavedModel(log + "/", ["autoencoder", "vae", "inverse", "forward"], cuda=True)
    srl2.model.eval()
    with torch.no_grad():
        st2 = srl2.model.getStates(obs.cuda()).cpu().numpy()
    assert np.abs(st2 - states[:8]).max() <= 1e-5 * np.abs(states[:8]).max()


def test_learn_stacked_split_model(workdir):
    """The reference's stacked-model configuration (tests/test_modules.py:8-19: `dae:1:20 reward:1:-1 forward:1:60
    inverse:5:20`, mlp inverse model, l2 regularisation, occlusion 0.3) through learn(), scaled to the tiny dataset."""
    from collections import OrderedDict
    import models.learner as learner
    import preprocessing.preprocess as pre
    from models.learner import SRL4robotics
    from models.modules import SRLModulesSplit
    from oracle import torch_twin as T
    name, paths, actions, rewards, starts = workdir
    pre.N_CHANNELS = 3
    learner.N_EPOCHS, learner.BATCH_SIZE, learner.VALIDATION_SIZE, learner.DISPLAY_PLOTS = 2, 8, 0.2, False
    log = "logs/run_split"
    os.makedirs(log, exist_ok=True)
    split = OrderedDict([("dae", 4), ("reward", -1), ("forward", 6), ("inverse", 2)])
    weights = OrderedDict([("dae", 1.0), ("reward", 1.0), ("forward", 1.0), ("inverse", 5.0)])
    srl = SRL4robotics(12, model_type="custom_cnn", inverse_model_type="mlp", seed=3, learning_rate=1e-3, cuda=True,
                       losses=list(split.keys()), losses_weights_dict=weights, n_actions=6, log_folder=log,
                       split_dimensions=split, l2_reg=1e-4, occlusion_percentage=0.3)
    assert isinstance(srl.model, SRLModulesSplit)
    loss_history, states, pairs = srl.learn(paths, actions, rewards, starts)
    assert states.shape == (len(paths), 12) and np.isfinite(states).all()
    assert [n for n, _ in pairs] == ["l2_loss", "forward_loss", "inverse_loss", "reward_loss", "reconstruction_loss"]
    assert dict(pairs)["inverse_loss"] == 5.0 and dict(pairs)["l2_loss"] == 1e-4
    for k in ("train_loss", "val_loss", "reward_loss", "l2_loss"):
        assert len(loss_history[k]) == 2 and np.isfinite(loss_history[k]).all(), k
    sd = torch.load(log + "/srl_model.pth", map_location="cpu")
    from preprocessing.data_loader import DataLoader
    obs = torch.cat([DataLoader._makeBatchElement(p) for p in paths[:8]], 0)
    ref = T.get_states(T.clone_state(sd, requires_grad=False), obs, "ae")
    err = np.abs(states[:8] - ref.numpy()).max() / np.abs(ref.numpy()).max()
    assert err < 1e-4, err
    with open(log + "/exp_config.json", "w") as f:
        json.dump(OrderedDict([("state-dim", 12), ("losses", list(split.keys())), ("n_actions", 6),
                               ("model-type", "custom_cnn"), ("inverse-model-type", "mlp"),
                               ("split-dimensions", split)]), f)
    srl2, cfg = SRL4robotics.loadSavedModel(log + "/", ["dae", "reward", "inverse", "forward"], cuda=True)
    assert isinstance(srl2.model, SRLModulesSplit)


def test_learn_vae_with_perceptual_loss(workdir):
    """`--losses vae perceptual --path-to-dae <srl_model.pth of a DAE run>` (reference train.py:120-125, learner.py:317-326):
    a DAE is trained first, then a VAE whose decoder is trained through the frozen denoiser's encoder."""
    import models.learner as learner
    import preprocessing.preprocess as pre
    from models.learner import SRL4robotics
    name, paths, actions, rewards, starts = workdir
    pre.N_CHANNELS = 3
    learner.N_EPOCHS, learner.BATCH_SIZE, learner.VALIDATION_SIZE, learner.DISPLAY_PLOTS = 1, 8, 0.2, False
    os.makedirs("logs/run_dae", exist_ok=True)
    os.makedirs("logs/run_percep", exist_ok=True)
    dae = SRL4robotics(12, model_type="custom_cnn", seed=4, learning_rate=1e-3, cuda=True, losses=["dae"], n_actions=6,
                       log_folder="logs/run_dae", occlusion_percentage=0.3)
    dae.learn(paths, actions, rewards, starts)
    srl = SRL4robotics(10, model_type="custom_cnn", seed=5, learning_rate=1e-3, cuda=True, losses=["vae", "perceptual"],
                       n_actions=6, log_folder="logs/run_percep", path_to_dae="logs/run_dae/srl_model.pth", state_dim_dae=12,
                       losses_weights_dict={"vae": 0.5e-6, "perceptual": 1e-3})
    before = [p.detach().clone() for p in srl.model.model.decoder_conv.parameters()]
    loss_history, states, pairs = srl.learn(paths, actions, rewards, starts)
    assert [n for n, _ in pairs] == ["kl_loss", "denoising perceptual similarity"]
    assert states.shape == (len(paths), 10) and np.isfinite(states).all()
    assert np.isfinite(loss_history["denoising perceptual similarity"]).all()
    assert all(not p.requires_grad for p in srl.denoiser.parameters()) and not srl.denoiser.training
    # the decoder only receives gradient through the denoiser: it must have moved
    after = list(srl.model.model.decoder_conv.parameters())
    assert any((a.detach() - b).abs().max().item() > 0 for a, b in zip(after, before))


@pytest.mark.parametrize("losses", [["autoencoder", "inverse", "forward", "reward"], ["vae"]])
def test_graph_mode_follows_eager(losses):
    """hipGraph replay of the step body (SRL4robotics._graphStep: forward + losses + backward + gradient delivery + Adam with
    a device-side step counter, captured once per variant) against the eager step, minibatch by minibatch — including a
    validation minibatch in the middle and, for the VAE, the noise drawn inside the graph."""
    import models.learner as learner
    import preprocessing.preprocess as pre
    from models.learner import SRL4robotics
    from losses.losses import LossManager
    import golden_util as gu
    pre.N_CHANNELS = 3
    B = 4
    learner.BATCH_SIZE = B

    def run(use_graph):
        srl = SRL4robotics(16, model_type="custom_cnn", seed=9, learning_rate=1e-3, cuda=True, losses=losses, n_actions=6,
                           log_folder="/tmp", l2_reg=1e-4 if "vae" not in losses else 0.0)
        srl._use_graph = use_graph
        lm = LossManager(srl.model, None)
        torch.manual_seed(123)
        trace = []
        for step in range(6):
            obs, nxt, act = gu.golden_inputs(B, 3, 6, seed=500 + step)
            rew = torch.from_numpy(gu.golden_rewards(B, seed=500 + step)[1]).cuda()
            loss = srl.trainStep(torch.from_numpy(obs).cuda(), torch.from_numpy(nxt).cuda(),
                                 torch.from_numpy(act).view(-1, 1).cuda(), lm, validation_mode=(step == 3), rewards_st=rew)
            trace.append([float(loss.detach())] + [float(l.detach()) for l in lm.losses])
            assert len(lm.names) == len(lm.losses) == len(lm.weights)
        torch.cuda.synchronize()
        return np.array(trace), srl.flat_params.flat.clone(), [b.clone() for b in srl.model.buffers()], srl.optimizer.steps()

    eager, p0, b0, t0 = run(False)
    graph, p1, b1, t1 = run(True)
    assert t0 == t1 == 5
    np.testing.assert_allclose(graph, eager, rtol=2e-5, atol=1e-7)
    assert (p1 - p0).abs().max().item() <= 2e-5 * p0.abs().max().item()
    for x, y in zip(b0, b1):
        assert (x.double() - y.double()).abs().max().item() <= 2e-5 * max(1.0, x.double().abs().max().item())


def test_train_cli_end_to_end(workdir):
    """The command line itself, the way the reference's own tests drive it (tests/test_modules.py, test_pipeline.py: run
    `python train.py ...`, expect exit code 0 and the output files): the reference's stacked-model flags on the tiny dataset."""
    import subprocess
    import sys
    name = workdir[0]
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    log = os.path.join(os.getcwd(), "logs", "cli_run")
    args = ["--no-display-plots", "--data-folder", name, "--epochs", "1", "--seed", "0", "--val-size", "0.2",
            "--state-dim", "10", "--model-type", "custom_cnn", "-bs", "8", "--log-folder", log,
            "--losses", "dae:1:2", "reward:1:-1", "forward:1:6", "inverse:5:2", "--inverse-model-type", "mlp",
            "--occlusion-percentage", "0.3", "--l2-reg", "0.0001"]
    env = dict(os.environ)
    proc = subprocess.run([sys.executable, os.path.join(repo, "srl-zoo_amd", "train.py")] + args, cwd=os.getcwd(), env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert proc.returncode == 0, proc.stdout.decode("utf-8", "replace")[-3000:]
    for f in ("srl_model.pth", "exp_config.json", "states_rewards.npz", "image_to_state.json"):
        assert os.path.exists(os.path.join(log, f)), f
    cfg = json.load(open(os.path.join(log, "exp_config.json")))
    assert cfg["losses"] == ["dae", "reward", "forward", "inverse"] and cfg["split-dimensions"]["forward"] == 6
    z = np.load(os.path.join(log, "states_rewards.npz"))
    assert z["states"].shape[1] == 10 and np.isfinite(z["states"]).all()


def test_train_cli_multi_view_vae(tmp_path):
    """`--multi-view --losses vae`: two stacked camera views -> 6-channel CNNVAE (the reference-valid half of BASELINE.json
    configs[4]; the reference itself crashes on `vae triplet`)."""
    import subprocess
    import sys
    make_dataset(str(tmp_path), name="tiny_mv", n_episodes=3, ep_len=14, multi_view=True)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    log = str(tmp_path / "logs" / "mv")
    args = ["--no-display-plots", "--data-folder", "tiny_mv", "--epochs", "1", "--state-dim", "6", "-bs", "4", "--multi-view",
            "--losses", "vae", "--log-folder", log]
    proc = subprocess.run([sys.executable, os.path.join(repo, "srl-zoo_amd", "train.py")] + args, cwd=str(tmp_path),
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert proc.returncode == 0, proc.stdout.decode("utf-8", "replace")[-3000:]
    sd = torch.load(os.path.join(log, "srl_model.pth"), map_location="cpu")
    assert tuple(sd["model.encoder_conv.0.weight"].shape) == (64, 6, 7, 7)
    assert tuple(sd["model.decoder_conv.12.weight"].shape) == (64, 6, 4, 4)
    z = np.load(os.path.join(log, "states_rewards.npz"))
    assert z["states"].shape[1] == 6 and np.isfinite(z["states"]).all()
