"""The multi-rank control flow of `train.py` / `SRL4robotics.learn()` EXECUTED with two ranks (reference
models/learner.py:285-292,501-522 is the single-process loop these ranks replicate): one process per rank started the way
torch.distributed.run starts them, the product's command line unmodified (tests/ddp_train_launcher.py only observes).

With >= 2 GPUs the ranks use backend "nccl" (RCCL over xGMI), one GPU each — the product configuration.  On a 1-GPU box
both ranks share GPU 0 and the process group is gloo (SRLZ_DIST_BACKEND=gloo: RCCL refuses two ranks on one device; the
bucket bounces through host memory, every kernel still runs on the GPU), so the test runs wherever one MI355X is visible.

Checked: ONE log folder (rank 0's timestamped choice, broadcast), identical loss_history / parameters on both ranks, lock-step
validation, the checkpoint's BatchNorm running statistics = the ranks' average, learn() returns on every rank and rank 0 writes
the reference's output files; the resident dataset completed by the ranks' slice exchange after epoch 1 (epoch 2 index-only on
both ranks); a NaN injected on rank 1 makes BOTH ranks exit with pipeline.NAN_ERROR (11) together.
"""
import glob
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from dataset_util import make_dataset

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAUNCHER = os.path.join(REPO, "tests", "ddp_train_launcher.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_ranks(world, args, cwd, extra_env, timeout=600):
    """Start `world` ranks of the launcher; returns [(returncode, combined output)] per rank."""
    port = _free_port()
    backend = "nccl" if torch.cuda.device_count() >= world else "gloo"
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   SRLZ_DIST_BACKEND=backend)
        env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, LAUNCHER] + args, cwd=cwd, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    out = []
    for p in procs:
        try:
            text, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            text, _ = p.communicate()
            text += b"\n[test] TIMEOUT"
        out.append((p.returncode, text.decode("utf-8", "replace")))
    return out, backend


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    root = tmp_path_factory.mktemp("ddp_learn")
    make_dataset(str(root), name="tiny_ddp", n_episodes=4, ep_len=26)
    return root


COMMON = ["--no-display-plots", "--data-folder", "tiny_ddp", "--epochs", "2", "--seed", "0", "--val-size", "0.2", "--state-dim", "10",
          "--model-type", "custom_cnn", "-bs", "8", "-lr", "0.001", "--losses", "autoencoder", "inverse", "forward"]


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
@pytest.mark.timeout(900)
def test_two_rank_train_cli(dataset):
    digest = dataset / "digest"
    digest.mkdir()
    results, backend = _run_ranks(2, COMMON, str(dataset), {"SRLZ_TEST_DIGEST_DIR": str(digest)})
    for rc, text in results:
        assert rc == 0, text[-4000:]
    ranks = [json.load(open(str(digest / ("rank%d.json" % r)))) for r in range(2)]
    assert [d["world"] for d in ranks] == [2, 2] and ranks[0]["backend"] == backend
    # ---- one log folder: rank 0's (timestamped) choice, used by both
    folders = glob.glob(str(dataset / "logs" / "tiny_ddp" / "*"))
    assert len(folders) == 1, folders
    assert ranks[0]["log_folder"] == ranks[1]["log_folder"]
    log = folders[0]
    assert os.path.samefile(log, os.path.join(str(dataset), ranks[0]["log_folder"]))
    # ---- identical histories and parameters (the scalar tail and the gradients travel in one bucket)
    assert ranks[0]["loss_history"] == ranks[1]["loss_history"]
    h = ranks[0]["loss_history"]
    assert set(h) >= {"train_loss", "val_loss", "reconstruction_loss", "inverse_loss", "forward_loss"}
    assert all(len(v) == 2 and np.isfinite(v).all() for v in h.values())
    assert ranks[0]["param_sum"] == ranks[1]["param_sum"] and ranks[0]["param_abs_sum"] == ranks[1]["param_abs_sum"]
    # 12 minibatches of 8, 2 for validation -> every rank: 5 optimisation steps and 1 validation step per epoch
    assert ranks[0]["adam_steps"] == ranks[1]["adam_steps"] == 10
    for d in ranks:
        assert d["states_shape"] == [104, 10] and d["states_finite"]
    # ---- the resident dataset is rank-aware: each rank decoded its slice beside epoch 1, the slices were exchanged at the epoch
    # boundary over the process group, and BOTH ranks train epoch 2 from indices only, from its first minibatch on
    for d in ranks:
        e1, e2 = d["epoch_stats"]
        assert d["resident_complete"]
        assert e1["minibatches"] == e2["minibatches"] == 6 and e1["index_minibatches"] == 0
        assert e2["index_minibatches"] == e2["minibatches"]
        assert e1["exchange"]["bytes"] == 104 * 3 * 224 * 224 and e1["exchange"]["backend"] == backend
    # ---- checkpoints: every save averaged the ranks' LOCAL running statistics (they differ: different minibatches)
    assert len(ranks[0]["saves"]) == len(ranks[1]["saves"]) >= 1
    for s0, s1 in zip(ranks[0]["saves"], ranks[1]["saves"]):
        assert s0["local_mean"] != s1["local_mean"]
        mean = (np.array(s0["local_mean"]) + np.array(s1["local_mean"])) / 2
        np.testing.assert_allclose(s0["avg_mean"], mean, rtol=1e-6, atol=1e-7)
        assert s0["avg_mean"] == s1["avg_mean"] and s0["tracked"] == s1["tracked"]
    sd = torch.load(os.path.join(log, "srl_model.pth"), map_location="cpu")
    saved = sd["model.encoder_conv.1.running_mean"].double().numpy()
    assert any(np.allclose(saved, s["avg_mean"], rtol=1e-6, atol=1e-7) for s in ranks[0]["saves"])
    # ---- rank 0 wrote the reference's output files
    for f in ("srl_model.pth", "exp_config.json", "states_rewards.npz", "image_to_state.json", "loss_history.npz"):
        assert os.path.exists(os.path.join(log, f)), f
    z = np.load(os.path.join(log, "loss_history.npz"))
    np.testing.assert_allclose(z["train_loss"], h["train_loss"])


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
@pytest.mark.timeout(900)
def test_nan_on_one_rank_exits_every_rank_with_11(dataset):
    results, _ = _run_ranks(2, COMMON + ["--log-folder", str(dataset / "logs" / "nan_run")], str(dataset), {"SRLZ_TEST_NAN_RANK": "1"})
    for rc, text in results:
        assert rc == 11, (rc, text[-3000:])
        assert "NaN Loss" in text


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
@pytest.mark.timeout(600)
def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts the two ranks itself and rank 0 prints ONE
    JSON line with n_gpus = rccl_ranks = 2 (on a 1-GPU box: both ranks on GPU 0 over gloo)."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env["SRLZ_DIST_BACKEND"] = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    proc = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                           "--batch-size", "8", "--timer-steps", "2", "--allow-short"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=500)
    assert proc.returncode == 0, proc.stderr.decode("utf-8", "replace")[-3000:]
    lines = [l for l in proc.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["rccl_ranks"] == 2 and out["config"]["global_batch"] == 16
    assert out["value"] > 0 and out["scaling"] == "weak" and "cpu_baseline" not in out
    assert np.isfinite(out["config"]["final_loss"])
    # what makes a first multi-GPU run readable: every rank's own step time, the all-reduce timed by HIP events, a strong-scaling leg
    assert len(out["ranks"]["ms_per_step_per_rank"]) == 2 and 0 < out["ranks"]["ms_per_step_min"] <= out["ranks"]["ms_per_step_max"]
    ar = out["allreduce"]
    assert ar["calls_timed"] == 2 and ar["avg_us"] > 0 and ar["bucket_bytes"] > 4 * 1e6 and ar["backend"] == env["SRLZ_DIST_BACKEND"]
    assert ar["busbw_GBps"] == pytest.approx(ar["algbw_GBps"], rel=1e-2)  # 2 (W - 1) / W = 1 at W = 2
    st = out["strong"]
    assert st["global_batch"] == 8 and st["per_gpu_batch"] == 4 and st["timed_region_s"] >= 0.2 and st["images_per_s"] > 0
    # the headline is both legs of the metric: the VAE step measured in the same process
    assert out["vae"]["ms_per_step"] > 0 and np.isfinite(out["vae"]["final_loss"]) and out["timed_region_s"] > 0


# ---------------------------------------------------------------------------------------------------------------------
# World 8 (round 6): the rank count of BASELINE.json configs[3] / [4] and of the driver's scaling run, on however many GPUs the box
# has — eight ranks on one MI355X in the gloo debug topology when it has one (every kernel still runs on the GPU; the bucket bounces
# through the host).  What a first run on an 8-GPU node exercises besides RCCL itself: eight training loaders + eight fill passes on
# one host (decoding threads budgeted against the usable cores, srlz.optim.loader_workers), the slice exchange among eight owners,
# lock-step step counts, one log folder, the fields of bench.py's multi-rank line.  With >= 8 GPUs the same tests run over RCCL.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def dataset8(tmp_path_factory):
    root = tmp_path_factory.mktemp("ddp_learn8")
    make_dataset(str(root), name="tiny_ddp8", n_episodes=4, ep_len=42)
    return root


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
@pytest.mark.timeout(1200)
def test_eight_rank_train_cli(dataset8):
    digest = dataset8 / "digest"
    digest.mkdir()
    args = ["--no-display-plots", "--data-folder", "tiny_ddp8", "--epochs", "2", "--seed", "0", "--val-size", "0.25", "--state-dim", "10",
            "--model-type", "custom_cnn", "-bs", "4", "-lr", "0.001", "--losses", "autoencoder", "inverse"]
    results, backend = _run_ranks(8, args, str(dataset8), {"SRLZ_TEST_DIGEST_DIR": str(digest)}, timeout=1000)
    for rc, text in results:
        assert rc == 0, text[-4000:]
    ranks = [json.load(open(str(digest / ("rank%d.json" % r)))) for r in range(8)]
    assert [d["world"] for d in ranks] == [8] * 8 and all(d["backend"] == backend for d in ranks)
    assert len(glob.glob(str(dataset8 / "logs" / "tiny_ddp8" / "*"))) == 1
    # identical histories, parameters and step counts on all eight ranks
    for d in ranks[1:]:
        assert d["loss_history"] == ranks[0]["loss_history"]
        assert d["param_sum"] == ranks[0]["param_sum"] and d["param_abs_sum"] == ranks[0]["param_abs_sum"]
        assert d["adam_steps"] == ranks[0]["adam_steps"] > 0
    assert all(np.isfinite(v).all() for v in ranks[0]["loss_history"].values())
    usable = ranks[0]["loader_placement"]["usable_cores"]
    for d in ranks:
        e1, e2 = d["epoch_stats"]
        # eight loaders + eight fill passes survived; the slices of eight owners were exchanged; epoch 2 is index-only everywhere
        assert d["resident_complete"] and e1["index_minibatches"] == 0 and e2["index_minibatches"] == e2["minibatches"] == e1["minibatches"]
        assert e1["exchange"]["bytes"] == 168 * 3 * 224 * 224
        assert d["states_shape"] == [168, 10] and d["states_finite"]
        # decoding threads x 2 loader processes x 8 local ranks (+ the 8 training threads) fit the host, or are down to one thread
        w = d["loader_placement"]["n_workers"]
        assert 1 <= w <= 4 and (w == 1 or w * 2 * 8 + 8 <= usable), d["loader_placement"]


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
@pytest.mark.timeout(900)
def test_bench_eight_ranks():
    """`python bench.py --gpus 8 --batch-size 8 --allow-short`: eight self-launched ranks, ONE JSON line with the multi-rank fields the
    driver's scaling run will be read through."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env["SRLZ_DIST_BACKEND"] = "nccl" if torch.cuda.device_count() >= 8 else "gloo"
    proc = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
                           "--batch-size", "8", "--timer-steps", "2", "--allow-short", "--no-vae-leg"], env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=800)
    assert proc.returncode == 0, proc.stderr.decode("utf-8", "replace")[-3000:]
    lines = [l for l in proc.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["rccl_ranks"] == 8 and out["config"]["global_batch"] == 64
    assert out["config"]["devices"] == (8 if env["SRLZ_DIST_BACKEND"] == "nccl" else torch.cuda.device_count())
    assert out["value"] > 0 and out["scaling"] == "weak" and np.isfinite(out["config"]["final_loss"])
    assert len(out["ranks"]["ms_per_step_per_rank"]) == 8
    ar = out["allreduce"]
    assert ar["calls_timed"] == 2 and ar["avg_us"] > 0 and ar["backend"] == env["SRLZ_DIST_BACKEND"]
    assert ar["busbw_GBps"] == pytest.approx(ar["algbw_GBps"] * 2 * 7 / 8, rel=1e-2)
    st = out["strong"]
    assert st["scaling"] == "strong" and st["global_batch"] == 8 and st["per_gpu_batch"] == 1 and st["images_per_s"] > 0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the library's own RCCL communicator needs one GPU per rank: runs the day the box has two")
@pytest.mark.timeout(900)
def test_bench_native_rccl_communicator():
    """SRLZ_COMM=rccl: the bucket through srlz_comm_allreduce_f32 (include/srlz.h) instead of torch.distributed, on as many ranks as the
    box has GPUs (at most 8).  Skipped on a 1-GPU lease — RCCL refuses two ranks on one device — and kept for the first node with more."""
    n = min(8, torch.cuda.device_count())
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(SRLZ_DIST_BACKEND="nccl", SRLZ_COMM="rccl")
    proc = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1",
                           "--batch-size", "8", "--timer-steps", "2", "--allow-short", "--no-vae-leg"], env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=800)
    assert proc.returncode == 0, proc.stderr.decode("utf-8", "replace")[-3000:]
    out = json.loads([l for l in proc.stdout.decode().splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == n and out["config"]["rccl_ranks"] == n and out["config"]["devices"] == n
    assert out["allreduce"]["backend"] == "nccl" and np.isfinite(out["config"]["final_loss"])
