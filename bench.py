#!/usr/bin/env python
"""bench.py — images/s of the srl-zoo image-representation TRAIN STEP on MI355X (BASELINE.json metric).

One "step" = one body of the reference's minibatch loop (models/learner.py:362-497) on one synthetic minibatch that
is already resident in HBM: forward of obs and next_obs through the conv auto-encoder, reconstruction loss, backward,
(one RCCL all-reduce of the flat gradient bucket when N > 1) and the fused Adam step.  Workload at N=1 =
BASELINE.json configs[1]: synthetic 224x224x3 observations, --losses autoencoder, custom_cnn, state-dim 200, bs=256.
`--losses vae` / `--losses autoencoder inverse forward` select configs[2] / configs[3]'s per-GPU workload.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 runs one process per GPU over RCCL.  Either a launcher provides the ranks (python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* are read from the environment), or — when
WORLD_SIZE is not set — bench.py starts the N ranks itself (self_launch) and rank 0 prints the line.

Prints ONE JSON line (rank 0).  value = 2*B*N*K / t  (both frames of a sample go through forward+backward), with t the
max over ranks of the wall time of exactly K steps bracketed by barrier + synchronize.
"roofline" is measured live with HIP events on the stream the kernels are launched on, for the MFMA implicit-GEMM convolution
kernel (conv64_fwd_kernel), in a short instrumented pass of the same steps RIGHT AFTER the timed region (two events per launch
inside the timed region cost ~0.5 % of the headline, VERDICT r2 item 11 — the timed region carries no instrumentation);
"cpu_baseline" times the CPU oracle (plain torch fp32 twin
of the reference path) on the host cores for a bounded sample — a reported baseline, not a target.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(REPO, "srl-zoo_amd"), REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def synthetic_batch(B, C, seed, device):
    """uint8 noise -> /255 -> ImageNet mean/std -> reference layout."""
    from golden_util import synthetic_obs
    obs, next_obs = synthetic_obs(B, C, seed)
    actions = np.random.RandomState(seed + 77).randint(0, 6, (B,)).astype(np.int64)
    # the two frames are the halves of ONE device buffer, as the learner's feed delivers them (SRL4robotics._toDevicePair)
    both = torch.from_numpy(np.concatenate((obs, next_obs), 0)).to(device)
    return both[:B], both[B:], torch.from_numpy(actions).view(-1, 1).to(device)


def host_cores():
    """CPUs this process may actually use: min(affinity, cgroup cpu.max quota) — the GPU box's container is capped."""
    from srlz.optim import usable_cores
    return usable_cores()


def cpu_baseline(losses, full=False):
    """The CPU oracle (oracle/torch_twin.py: the reference's own torch CPU ops, fp32, pinned to the reference by the golden
    fixtures) timed on this box's host cores with BASELINE.md section 3's protocol: the same train step (forward x2, losses,
    backward, Adam); bs = 32 (BASELINE.json configs[0]'s minibatch) with all usable cores, 3 warm-up + 10 timed steps, median —
    that is `value`; plus a bs = 256 leg (configs[1]'s minibatch).  Default run: the bs = 256 leg is BOUNDED to 1 warm-up + 2
    timed steps (a step takes seconds; the whole leg stays within ~40 s of CPU work) and the 8-thread comparability leg is
    skipped; `--cpu-baseline-full` runs every leg of the protocol (3 + 10 steps at bs = 32 with all cores and with 8 threads,
    3 + 10 at bs = 256: minutes).  kind = "port": a restatement of the reference path, not the reference's own files (those
    cannot travel to the GPU box)."""
    from oracle import torch_twin as T
    import preprocessing.preprocess as pre
    from models.modules import SRLModules
    from golden_util import synthetic_obs
    cores = host_cores()
    pre.N_CHANNELS = 3

    def run(bs, threads, warm, steps):
        torch.set_num_threads(threads)
        np.random.seed(1)
        torch.manual_seed(1)
        model = SRLModules(state_dim=200, action_dim=6, cuda=False, model_type="custom_cnn", losses=losses)
        sd = T.clone_state(model.state_dict())
        opt = T.TwinAdam(sd, 0.005)
        obs, next_obs = synthetic_obs(bs, 3, 4321)
        obs, next_obs = torch.from_numpy(obs), torch.from_numpy(next_obs)
        actions = torch.randint(0, 6, (bs,))
        eps = [torch.randn(bs, 200), torch.randn(bs, 200)] if "vae" in losses else [None, None]
        times = []
        for i in range(warm + steps):
            t0 = time.time()
            T.train_step(sd, losses, obs, next_obs, actions, eps=eps[0], next_eps=eps[1])
            opt.step(sd)
            if i >= warm:
                times.append(time.time() - t0)
        med = float(np.median(times))
        return {"bs": bs, "threads": threads, "warmup": warm, "steps": steps, "s_per_step": round(med, 3),
                "images_per_s": round(2 * bs / med, 2), "samples_per_s": round(bs / med, 2)}

    runs = [run(32, cores, 3, 10)]
    if full:
        runs += [run(32, min(8, cores), 3, 10), run(256, cores, 3, 10)]
    else:
        runs += [run(256, cores, 1, 2)]
    torch.set_num_threads(cores)
    head = runs[0]
    return {"value": head["images_per_s"], "unit": "images/s", "cores": head["threads"], "kind": "port",
            "protocol": "BASELINE.md section 3" + ("" if full else " (bs=256 leg bounded to 1+2 steps, 8-thread leg skipped: "
                                                              "--cpu-baseline-full runs them)"),
            "sample": "median of %d train steps (fwd x2, losses, bwd, Adam) at bs=32 (64 images each) after %d warm-ups; torch %s CPU "
                      "fp32, %d threads = all usable cores; %.2f s/step" % (head["steps"], head["warmup"], torch.__version__,
                                                                            head["threads"], head["s_per_step"]),
            "runs": runs}


def conv64_algorithmic_bytes(layer_key):
    """Input + output activations (fp32 NHWC, 64 channels) + the 9x64x64 weights of one launch; key = ops._conv64_key
    (n = images per launch: 2 x bs when the two frames of a step are batched)."""
    import re
    m = re.search(r" n(\d+) (\d+)x(\d+)->(\d+)x(\d+)", layer_key)
    n, hi, wi, ho, wo = (int(g) for g in m.groups())
    return 4.0 * 64 * n * (hi * wi + ho * wo) + 4.0 * 9 * 64 * 64


def csrc_sha16():
    """Fingerprint of the kernel sources (csrc/*.hip, *.cpp, *.h): profiles/*_pmc_*.json carry the value they were recorded at."""
    import glob
    import hashlib
    h = hashlib.sha256()
    root = os.path.join(REPO, "srl-zoo_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.cpp")) + glob.glob(os.path.join(root, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _newest_profile(suffix):
    """The committed profile file to quote: the one recorded at THESE kernel sources (csrc_sha16) when there is one — tags do not sort by
    time (r06w was recorded after r06z) — else the last by name (the caller then reports stale = True)."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*" + suffix)))
    if not files:
        return None, None
    sha = csrc_sha16()
    docs = []
    for f in reversed(files):
        try:
            doc = json.load(open(f))
        except (OSError, ValueError):
            continue
        if doc.get("csrc_sha16") == sha:
            return doc, "profiles/" + os.path.basename(f)
        docs.append((doc, f))
    if not docs:
        return None, None
    return docs[0][0], "profiles/" + os.path.basename(docs[0][1])


def committed_pmc_traffic(kernel):
    """(HBM bytes per launch of `kernel`, source file, stale) from the newest profiles/*_pmc_traffic.json (rocprofv3 PMC passes
    of this same command, collected by tools/profile_round.sh and committed; counters cannot be read from inside the process).
    stale = the file was recorded at other kernel sources than the ones this run was built from (or carries no fingerprint)."""
    doc, src = _newest_profile("_pmc_traffic.json")
    rec = (doc or {}).get("kernels", {}).get(kernel)
    if not rec:
        return None, None, None
    return rec["hbm_bytes_per_launch"], src, doc.get("csrc_sha16") != csrc_sha16()


def committed_kernel_trace(kernel):
    """(average launch duration in us of `kernel`, source file, stale) from the rocprofv3 --kernel-trace --stats summary of this same
    command that tools/profile_round.sh recorded next to the newest PMC passes (profiles/<tag>_bench_ae_bs256_kernel_stats.csv):
    the file whose average the live HIP-event time of the roofline object must agree with.  stale as in committed_pmc_traffic
    (the CSV carries no fingerprint of its own: its tag's PMC file does)."""
    import csv
    doc, src = _newest_profile("_pmc_traffic.json")
    if not src:
        return None, None, None
    path = os.path.join(REPO, src.replace("_pmc_traffic.json", "_bench_ae_bs256_kernel_stats.csv"))
    if not os.path.exists(path):
        return None, None, None
    try:
        for row in csv.DictReader(open(path)):
            name = row["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
            if name.startswith(kernel + "(") or name.startswith(kernel + "<"):
                return round(float(row["AverageNs"]) / 1e3, 2), "profiles/" + os.path.basename(path), doc.get("csrc_sha16") != csrc_sha16()
    except (OSError, ValueError, KeyError):
        pass
    return None, "profiles/" + os.path.basename(path), doc.get("csrc_sha16") != csrc_sha16()


def committed_pmc_mfma(kernel):
    """(matrix-pipe busy fraction, measured clock, source file, stale) of `kernel` from the newest profiles/*_pmc_mfma.json."""
    doc, src = _newest_profile("_pmc_mfma.json")
    rec = (doc or {}).get("kernels", {}).get(kernel)
    if not rec:
        return None, None, None, None
    return rec["mfma_busy_frac"], rec["clock_ghz"], src, doc.get("csrc_sha16") != csrc_sha16()


def rank_environments(n, port, base=None):
    """Environment of each of the n ranks bench.py starts itself (one process per GPU, rendezvous on 127.0.0.1) — the same
    variables torch.distributed.run would set."""
    base = dict(os.environ if base is None else base)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
    envs = []
    for r in range(n):
        e = dict(base)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=str(port))
        envs.append(e)
    return envs


def self_launch(n, argv, script=None):
    """`python bench.py --gpus N` without a launcher: start the N ranks (this same command line, one process per GPU), let
    rank 0 own stdout (the ONE JSON line), wait for all; a failing rank takes the others down.  Returns the exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r, env in enumerate(rank_environments(n, port)):
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:  # the others would hang in their next collective
                    q.terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-size", type=int, default=256, help="per-GPU minibatch (samples; 2 frames each)")
    ap.add_argument("--losses", nargs="+", default=["autoencoder"])
    ap.add_argument("--state-dim", type=int, default=200)
    ap.add_argument("--channels", type=int, default=3, choices=[3, 6, 9],
                    help="input channels: 6 = --multi-view (two stacked cameras, BASELINE.json configs[4] `vae` half); "
                         "`--losses triplet` forces 9 (anchor / positive / negative views, configs[4] `triplet` half)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="every leg of BASELINE.md section 3's CPU protocol (minutes) instead of the bounded default")
    ap.add_argument("--timer-steps", type=int, default=10,
                    help="instrumented steps (HIP events around every MFMA launch) run AFTER the timed region for the roofline object")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--host-input", action="store_true",
                    help="NOT the contract's metric: every step starts from uint8 frames [B,C,W,H] in pinned HOST memory (the H2D "
                         "copy inside the timed region; conv1 and the fused loss read the bytes) — the PCIe-inclusive rate "
                         "quoted in DESIGN.md")
    ap.add_argument("--u8-resident", action="store_true",
                    help="the resident synthetic batch is the loader's uint8 frames [B,C,W,H] instead of the normalised float tensor "
                         "(what learn() feeds the step; conv1 and the fused loss read the bytes)")
    ap.add_argument("--no-vae-leg", action="store_true",
                    help="skip the second leg of the headline metric (BASELINE.json configs[2]: --losses vae on the same shapes), which the "
                         "default --losses autoencoder run measures in the same process right after the AE leg and reports as `vae`")
    ap.add_argument("--no-strong-leg", action="store_true",
                    help="with --gpus N > 1: skip the strong-scaling leg (global batch = --batch-size, i.e. bs / N per GPU) reported as `strong`")
    ap.add_argument("--allow-short", action="store_true",
                    help="print the line even when the timed region is shorter than 0.1 s (functional tests on tiny batches)")
    ap.add_argument("--host-input-nhwc", action="store_true",
                    help="with --host-input: frames as decoded ([B,H,W,C]) + the separate srlz_normalize_u8 pass (A/B)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))  # no launcher: start the ranks ourselves
    if args.gpus != world:
        raise SystemExit("--gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    from srlz import optim as _optim
    local_dev = _optim.local_device_index() if world > 1 else local_rank
    torch.cuda.set_device(local_dev)
    device = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group(backend=_optim.dist_backend())  # "nccl" = RCCL over xGMI
        # (collective) the communicator's own view of the job: under RCCL one distinct GPU per rank, or every rank stops with a message
        placement = _optim.rank_devices()
        # communicator set-up (seconds) must never land in the timed region, whatever --warmup says
        warm = torch.zeros(1 << 20, device=device)
        _optim._sum_across_ranks(warm)
        torch.cuda.synchronize()
        if _optim.native_comm_requested():  # SRLZ_COMM=rccl: the library's own RCCL communicator (include/srlz.h)
            _optim.init_native_comm()
            _optim._sum_across_ranks(warm)
            torch.cuda.synchronize()

    import models.learner as learner
    from models.learner import SRL4robotics
    from losses.losses import LossManager
    from srlz import ops

    B = args.batch_size
    learner.BATCH_SIZE = B
    import preprocessing.preprocess as pre
    channels = 9 if "triplet" in args.losses else args.channels
    pre.N_CHANNELS = channels
    quiet = open(os.devnull, "w")
    stdout, sys.stdout = sys.stdout, quiet  # the learner prints its banner; keep stdout to the JSON line
    try:
        srl = SRL4robotics(args.state_dim, model_type="custom_cnn", seed=1, learning_rate=0.005, cuda=True,
                           losses=list(args.losses), n_actions=6, beta=1.0, log_folder="/tmp", multi_view=channels > 3)
    finally:
        sys.stdout = stdout
    loss_manager = LossManager(srl.model, None)
    obs, next_obs, actions = synthetic_batch(B, channels, 1234 + rank, device)
    rewards = None
    if "reward" in args.losses:
        rewards = torch.from_numpy(np.random.RandomState(99 + rank).randint(0, 2, (B,)).astype(np.int64)).to(device)

    if args.u8_resident:
        rs = np.random.RandomState(4321 + rank)
        both = torch.from_numpy(rs.randint(0, 256, (2 * B, channels, 224, 224)).astype(np.uint8)).to(device)
        obs, next_obs = srl._toDevicePair(both[:B], both[B:])
    host_frames = None
    if args.host_input or args.host_input_nhwc:
        # what the loader hands over every step (SRL4robotics.learn(): DataLoader(raw_uint8="planar")): uint8 frames [B,C,W,H] in
        # pinned host memory.  As in learn() (_DeviceFeed), the NEXT step's frames cross PCIe on a copy stream while this step
        # computes and land as the halves of one device buffer; conv1 and the fused reconstruction loss read the bytes
        # (srlz_conv1_fwd_u8 ...).  --host-input-nhwc: the frames as decoded ([B,H,W,C]) + srlz_normalize_u8 (the round-2 route).
        rs = np.random.RandomState(4321 + rank)
        fshape = (B, 224, 224, channels) if args.host_input_nhwc else (B, channels, 224, 224)
        srl.frame_layout = "nhwc" if args.host_input_nhwc else "planar"  # (stated, not guessed: BaseLearner.frame_layout)
        host_frames = [[torch.from_numpy(rs.randint(0, 256, fshape).astype(np.uint8)).pin_memory()
                        for _ in range(2)] for _ in range(2)]  # two alternating minibatches
        copy_stream = torch.cuda.Stream(device=device)
        ahead = {}

        def upload(i):
            with torch.cuda.stream(copy_stream):
                both = torch.empty((2 * B,) + fshape[1:], dtype=torch.uint8, device=device)
                both[:B].copy_(host_frames[i % 2][0], non_blocking=True)
                both[B:].copy_(host_frames[i % 2][1], non_blocking=True)
                dev = [both[:B], both[B:]]
            ev = torch.cuda.Event()
            ev.record(copy_stream)
            ahead[i] = (dev, ev)
        upload(0)
    counter = [0]

    def step():
        if host_frames is None:
            return srl.trainStep(obs, next_obs, actions, loss_manager, rewards_st=rewards)
        i = counter[0]
        counter[0] += 1
        dev, ev = ahead.pop(i)
        cur = torch.cuda.current_stream(device)
        cur.wait_event(ev)
        for t in dev:
            t.record_stream(cur)
        upload(i + 1)  # next minibatch travels while this step runs
        o, no = srl._toDevicePair(dev[0], dev[1])  # the bytes themselves when the step reads bytes, else normalised fp32
        return srl.trainStep(o, no, actions, loss_manager, rewards_st=rewards)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(step_fn, warmup, steps):
        """`warmup` untimed steps, then EXACTLY `steps` steps bracketed by barrier + synchronize.  Returns (seconds: max over the
        ranks, seconds on this rank's own clock, the steps' total losses)."""
        for _ in range(warmup):
            step_fn()
        sync()
        t0 = time.time()
        totals = [step_fn().detach() for _ in range(steps)]
        sync()
        own = time.time() - t0
        worst = own
        if world > 1:
            t = torch.tensor([own], dtype=torch.float64)
            if torch.distributed.get_backend() != "gloo":
                t = t.to(device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            worst = float(t.item())
        return worst, own, torch.stack(totals).tolist()

    dt, own_dt, last_losses = timed(step, args.warmup, args.steps)
    if dt < 0.1 and not args.allow_short:  # (0.2 s until round 6: twenty steps of the bs = 256 step are 0.23 s by now)
        # (dt is the maximum over the ranks, the same number everywhere: every rank takes this exit, here, before the other legs — a
        # rank 0 leaving alone would strand the others in their next collective until the RCCL timeout)
        if world > 1:
            torch.distributed.destroy_process_group()
        msg = "bench.py: the timed region was %.3f s (< 0.1 s: %d steps of %.3f ms) — too short to be a measurement; " \
              "raise --steps (or pass --allow-short for a functional check)" % (dt, args.steps, 1e3 * dt / args.steps)
        raise SystemExit(msg if rank == 0 else 1)
    rank_ms = None
    if world > 1:  # every rank's own clock around the same K steps: skew between the ranks shows here
        rank_ms = [None] * world
        torch.distributed.all_gather_object(rank_ms, round(1e3 * own_dt / args.steps, 3))
    # the roofline's per-launch HIP-event times: the SAME steps, instrumented, after the clock has stopped
    # (two instrumented steps are discarded first: the first ~200 event pairs are created cold and their launches measured
    # 3-4 % long — 448 vs 432 us for the dominant kernel against rocprofv3's 434)
    timer_steps = 0 if args.no_kernel_timers else max(0, args.timer_steps)
    for phase, count in (("settle", 2 if timer_steps else 0), ("measure", timer_steps)):
        if count and rank == 0:
            ops.timers_enable(True)  # (clears what the settling steps recorded)
        if count and world > 1:
            _optim.comm_timing(True)
        for _ in range(count):  # (every rank runs them: the steps hold collectives)
            step()
        sync()
    ops.timers_enable(False)
    comm = None
    if world > 1:
        # the step's ONE collective, timed with HIP events on the launch stream in the instrumented steps (every rank; rank 0 reports
        # its own average and the fastest / slowest rank's)
        mine = _optim.comm_timing_report()
        _optim.comm_timing(False)
        avg_us = float(np.mean([u for u, _ in mine])) if mine else None
        per_rank = [None] * world
        torch.distributed.all_gather_object(per_rank, avg_us)
        if mine:
            nbytes = mine[0][1]
            comm = {"collective": "all_reduce(sum) of the flat gradient bucket (gradients + 16 loss scalars), one per step",
                    "backend": torch.distributed.get_backend(), "native_rccl_comm": bool(_optim.native_comm_requested()),
                    "bucket_bytes": nbytes, "calls_timed": len(mine), "avg_us": round(avg_us, 1),
                    "avg_us_per_rank_min_max": [round(min(per_rank), 1), round(max(per_rank), 1)],
                    "algbw_GBps": round(nbytes / (avg_us * 1e-6) / 1e9, 2),
                    "busbw_GBps": round(nbytes / (avg_us * 1e-6) / 1e9 * 2.0 * (world - 1) / world, 2),
                    "share_of_step": round(avg_us * 1e-3 / (1e3 * dt / args.steps), 4),
                    "timing": "HIP events on the launch stream around the collective, %d instrumented steps" % timer_steps}

    # ---- the other leg of the headline metric ("AE+VAE train step"): BASELINE.json configs[2], --losses vae (beta = 1) on the same
    # shapes and batch, measured in this process right after the AE leg with the same warm-up / step counts
    vae = None
    if (list(args.losses) == ["autoencoder"] and channels == 3 and host_frames is None and not args.u8_resident
            and not args.no_vae_leg):
        stdout, sys.stdout = sys.stdout, quiet
        try:
            srl_vae = SRL4robotics(args.state_dim, model_type="custom_cnn", seed=1, learning_rate=0.005, cuda=True, losses=["vae"],
                                   n_actions=6, beta=1.0, log_folder="/tmp")
        finally:
            sys.stdout = stdout
        lm_vae = LossManager(srl_vae.model, None)
        v_dt, _, v_losses = timed(lambda: srl_vae.trainStep(obs, next_obs, actions, lm_vae), args.warmup, args.steps)
        v_tf = 2320.0e6 * 2 * B / (v_dt / args.steps) / 1e12
        vae = {"workload": "synthetic 224x224x3 obs, --losses vae (beta=1), custom_cnn, state-dim %d, bs=%d per GPU" % (args.state_dim, B),
               "ms_per_step": round(1e3 * v_dt / args.steps, 3), "images_per_s": round(2 * B * world * args.steps / v_dt, 1),
               "steps": args.steps, "warmup": args.warmup, "timed_region_s": round(v_dt, 4), "final_loss": round(v_losses[-1], 6),
               "step_roofline": {"algorithmic_tflop_per_step": round(2320.0e6 * 2 * B / 1e12, 4), "achieved_tflops": round(v_tf, 2),
                                 "frac_of_fp32_mfma_peak": round(v_tf / PEAK_FP32_MFMA_TFLOPS, 4)}}
        del srl_vae, lm_vae

    # ---- strong scaling beside weak (N > 1): the SAME global batch as the 1-GPU run, bs / N samples per GPU
    strong = None
    if world > 1 and B % world == 0 and host_frames is None and not args.no_strong_leg:
        b = B // world
        pair = torch.cat((obs[:b], next_obs[:b]), 0)  # the two frames as the halves of one buffer, like the full batch
        s_obs, s_next, s_act = pair[:b], pair[b:], actions[:b].contiguous()
        s_rew = None if rewards is None else rewards[:b].contiguous()

        def s_step():
            return srl.trainStep(s_obs, s_next, s_act, loss_manager, rewards_st=s_rew)
        probe, _, _ = timed(s_step, 3, 3)
        k = max(args.steps, int(np.ceil(0.25 / max(probe / 3, 1e-6))))
        if world > 1:  # (the step count must be the same on every rank)
            box = [k]
            torch.distributed.broadcast_object_list(box, src=0)
            k = box[0]
        s_dt, _, _ = timed(s_step, 0, k)
        while s_dt < 0.2 and k < (1 << 20):  # (the probe ran slower than the steps do: s_dt is the maximum over the ranks, so every
            k *= 2                           # rank takes the same decision)
            s_dt, _, _ = timed(s_step, 0, k)
        strong = {"scaling": "strong", "global_batch": B, "per_gpu_batch": b, "steps": k, "ms_per_step": round(1e3 * s_dt / k, 3),
                  "images_per_s": round(2 * B * k / s_dt, 1), "timed_region_s": round(s_dt, 4)}

    if rank == 0:
        images = 2 * B * world * args.steps
        if host_frames is not None:
            where = "uint8 frames [B,%s] in pinned host memory every step, H2D copy inside the timed region (PCIe-inclusive)" % (
                "H,W,C" if args.host_input_nhwc else "C,W,H")
        elif args.u8_resident:
            where = "the loader's uint8 frames [B,C,W,H] resident in HBM"
        else:
            where = "normalised fp32 observations resident in HBM"
        out = {
            "metric": "images/sec (224x224x3) AE+VAE train step", "value": round(images / dt, 1), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "timed_region_s": round(dt, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": ("synthetic, uint8 frames resident in HBM" if args.u8_resident else "synthetic") if host_frames is None else "synthetic, uint8 frames in pinned host memory every step (PCIe-inclusive)",
            "samples_per_s": round(B * world * args.steps / dt, 1),
            "value_is": "the --losses %s leg (BASELINE.json configs[1] when autoencoder at bs=256)%s" % (
                " ".join(args.losses), "; the VAE leg of the metric (configs[2]) is the `vae` object, same process, same step counts"
                if vae is not None else ""),
            "config": {"workload": "synthetic 224x224x%d obs, --losses %s, custom_cnn, state-dim %d, bs=%d per GPU "
                                   "(%d frames fwd+bwd per step per GPU), Adam lr 0.005, %s"
                                   % (channels, " ".join(args.losses), args.state_dim, B, 2 * B, where),
                       "global_batch": B * world, "parallelism": "dp%d" % world,
                       # (from the process group and its gathered device table, not from the environment)
                       "rccl_ranks": placement["ranks"] if world > 1 else 1,
                       "devices": placement["devices"] if world > 1 else 1,
                       "final_loss": round(last_losses[-1], 6)},
        }
        if vae is not None:
            out["vae"] = vae
        if world > 1:
            out["ranks"] = {"ms_per_step_per_rank": rank_ms, "ms_per_step_min": min(rank_ms), "ms_per_step_max": max(rank_ms),
                            "backend": torch.distributed.get_backend(),
                            "note": "each rank's own clock around the same K steps (barrier + synchronize on both sides)"}
            if comm is not None:
                out["allreduce"] = comm
            if strong is not None:
                out["strong"] = strong
        # whole-step arithmetic intensity check (2317.3 MFLOP per 224x224x3 image and train step for the AE,
        # 2320.0 for the VAE; heads are negligible): algorithmic FLOP of the step / wall time, against the same fp32-matrix peak
        if channels == 3 and "triplet" not in args.losses:
            per_image = 2320.0e6 if "vae" in args.losses else 2317.3e6
            step_tf = per_image * 2 * B / (dt / args.steps) / 1e12
            out["step_roofline"] = {"algorithmic_tflop_per_step": round(per_image * 2 * B / 1e12, 4),
                                    "achieved_tflops": round(step_tf, 2), "frac_of_fp32_mfma_peak": round(step_tf / PEAK_FP32_MFMA_TFLOPS, 4)}
        rep = ops.timers_report()
        # every instrumented kernel symbol (the MFMA launches of the step): launches, average, achieved rate, share of their time
        SYMBOL = {"conv64_fwd_kernel": "conv64_fwd_kernel<4, false>", "conv64_bwd_fused_kernel": "conv64_bwd_fused_kernel",
                  "conv64_dgrad_poolsum_kernel": "conv64_dgrad_poolsum_kernel<1>",
                  "conv64_fwd_kernel<bn-bwd operand>": "conv64_fwd_kernel<4, true>", "conv64_gather_pipe_kernel": "conv64_gather_pipe_kernel"}
        NOTE = {"conv64_fwd_kernel": "3x3 64->64 ConvTranspose forward (ConvT1-4), 4 launches per step (conv2: conv64_wino_kernel; conv3 forward and ConvT1 data gradient: conv64_gather_pipe_kernel)",
                "conv64_wino_kernel": "conv2 forward and data gradient as Winograd F(2x2,3x3); flop = the layer's direct-convolution flop",
                "conv64_wino_wgrad_kernel": "conv2 weight gradient by the transposed Winograd algorithm (+ two reduction launches)",
                "conv64_bwd_fused_kernel": "data + weight + bias gradient of a ConvTranspose block in one launch (ConvT2-4); flop = 2 x the "
                                           "layer's forward flop",
                "conv64_dgrad_poolsum_kernel": "conv2 / conv3 data gradient with the pooled block's BatchNorm-backward sums in its epilogue"}
        symbols = {name: v for name, v in rep.items() if "/" not in name and v["ms"] > 0}
        total_ms = sum(v["ms"] for v in symbols.values())
        by_symbol = {}
        for name, v in sorted(symbols.items(), key=lambda kv: -kv[1]["ms"]):
            tf = v["flop"] / (v["ms"] * 1e-3) / 1e12 if v["flop"] else None
            by_symbol[name] = {"launches": v["launches"], "avg_us": round(1e3 * v["ms"] / v["launches"], 2),
                               "share_of_instrumented_time": round(v["ms"] / total_ms, 4),
                               "tflops": None if tf is None else round(tf, 2),
                               "frac": None if tf is None else round(tf / PEAK_FP32_MFMA_TFLOPS, 4)}
            if tf is not None and name.startswith("conv64_wino"):  # (algorithmic FLOP of the layer; the kernel executes 16/36 of them)
                by_symbol[name]["executed_frac"] = round(16.0 / 36.0 * tf / PEAK_FP32_MFMA_TFLOPS, 4)
        with_flop = [n for n in by_symbol if by_symbol[n]["tflops"] is not None]
        dom = with_flop[0] if with_flop else None  # (sorted by time: the dominant MFMA kernel symbol of the step)
        k = rep.get(dom) if dom else None
        if k and k["ms"] > 0:
            tf = k["flop"] / (k["ms"] * 1e-3) / 1e12
            layers = {}
            for name, v in sorted(rep.items()):
                if name.startswith(dom + "/") and v["ms"] > 0:
                    layers[name.split("/", 1)[1]] = {"launches": v["launches"], "avg_us": round(1e3 * v["ms"] / v["launches"], 2),
                                                     "tflops": round(v["flop"] / (v["ms"] * 1e-3) / 1e12, 2)}
            alg_bytes = 0.0
            for key, v in layers.items():
                # forward / data gradient: input + output + weights; the fused block backward reads dA and y (output-sized), y_prev
                # (input-sized) and writes dA_prev (input-sized)
                b = conv64_algorithmic_bytes(key)
                if dom == "conv64_bwd_fused_kernel":
                    b = 2.0 * b - 4.0 * 9 * 64 * 64
                alg_bytes += v["launches"] * b
            alg_bytes /= max(1, k["launches"])
            sym = SYMBOL.get(dom, dom)
            traffic, traffic_src, traffic_stale = committed_pmc_traffic(sym)
            busy, clock, busy_src, busy_stale = committed_pmc_mfma(sym)
            trace_us, trace_src, trace_stale = committed_kernel_trace(sym)
            out["roofline"] = {"kernel": "%s (%s)" % (sym, NOTE.get(dom, "")),
                               "bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
                               "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic_src,
                               "algorithmic_bytes_per_launch": round(alg_bytes),
                               "mfma_busy_frac": busy, "clock_ghz": clock, "mfma_pmc_source": busy_src,
                               "kernel_trace_source": trace_src, "kernel_trace_avg_us": trace_us,
                               "stale": bool(traffic_stale or busy_stale or trace_stale), "csrc_sha16": csrc_sha16(),
                               "stale_note": "traffic / mfma_busy_frac / clock_ghz / kernel_trace_avg_us come from committed rocprofv3 passes of "
                                             "this command (tools/profile_round.sh); stale = recorded at other kernel sources than this build's",
                               "timing": "%d instrumented steps (after 2 discarded ones) behind the timed region (HIP events on the launch stream)" % timer_steps,
                               "launches": k["launches"], "avg_launch_us": round(1e3 * k["ms"] / k["launches"], 2),
                               "algorithmic_gflop_per_launch": round(k["flop"] / k["launches"] / 1e9, 3),
                               "layers": layers, "by_symbol": by_symbol}
        # BASELINE.json north_star: ">= 70 % of the CDNA4 fp32 MFMA roofline on the 3x3 conv encoder" = conv2 (3x3 s1, 56x56) and
        # conv3 (3x3 s2, 27x27 -> 14x14), reference models/models.py:54-62 — forward, data gradient and weight gradient of each,
        # from the same instrumented pass as `roofline` (whatever kernel symbol a launch ran under), and their FLOP-weighted aggregate
        ns, tot = {}, {"all": [0.0, 0.0], "conv2": [0.0, 0.0], "conv3": [0.0, 0.0]}  # [flop, ms]
        tot_ex = {"all": [0.0, 0.0], "conv2": [0.0, 0.0], "conv3": [0.0, 0.0]}       # ... with the multiply-adds actually executed
        for name, v in sorted(rep.items()):
            if "/" not in name or v["ms"] <= 0 or not v["flop"]:
                continue
            symbol, key = name.split("/", 1)
            layer = "conv2" if key.startswith("conv s1 ") else "conv3" if key.startswith("conv s2 ") else None
            what = key.rsplit(" ", 1)[1]
            if layer is None or what not in ("fwd", "dgrad", "wgrad"):
                continue
            tf = v["flop"] / (v["ms"] * 1e-3) / 1e12
            # Winograd F(2x2, 3x3) launches (csrc/wino.hip) execute 16 of the direct algorithm's 36 multiply-adds per (ci, co) and 2x2
            # patch: `tflops` / `frac` price the ALGORITHMIC work of the layer (what the reference's layer costs done
            # directly) and may exceed the matrix peak; `executed_frac` is what the matrix pipe actually ran
            wino = symbol.startswith("conv64_wino")
            ex = 16.0 / 36.0 if wino else 1.0
            ns["%s_%s" % (layer, what)] = {"kernel": symbol, "launches": v["launches"], "avg_us": round(1e3 * v["ms"] / v["launches"], 2),
                                           "gflop_per_launch": round(v["flop"] / v["launches"] / 1e9, 3),
                                           "tflops": round(tf, 2), "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4),
                                           "algorithm": "winograd F(2x2,3x3): 16/36 of the direct multiply-adds" if wino else "direct implicit GEMM",
                                           "executed_frac": round(ex * tf / PEAK_FP32_MFMA_TFLOPS, 4)}
            for t in (tot["all"], tot[layer]):
                t[0] += v["flop"]
                t[1] += v["ms"]
            for t in (tot_ex["all"], tot_ex[layer]):
                t[0] += ex * v["flop"]
                t[1] += v["ms"]
        if ns and channels == 3 and "triplet" not in args.losses:
            frac = {k: round(f / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) for k, (f, ms) in tot.items() if ms > 0}
            frac_ex = {k: round(f / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) for k, (f, ms) in tot_ex.items() if ms > 0}
            out["north_star"] = {"what": "the 3x3 conv encoder (conv2 3x3 s1 64->64 @56x56, conv3 3x3 s2 64->64 27x27->14x14): forward, data "
                                         "gradient, weight gradient; FLOP-weighted aggregate vs the fp32 MFMA peak (target >= 0.70).  "
                                         "aggregate_frac prices the layers' ALGORITHMIC (direct-convolution) FLOP, as every round before; since "
                                         "round 6 conv2 runs as Winograd F(2x2,3x3) (16/36 of those multiply-adds), so it can exceed 1 — "
                                         "aggregate_executed_frac is the matrix pipe's own utilisation over the same launches",
                                 "peak_tflops": PEAK_FP32_MFMA_TFLOPS, "launch": ns, "frac_conv2": frac.get("conv2"),
                                 "frac_conv3": frac.get("conv3"), "aggregate_frac": frac.get("all"),
                                 "aggregate_executed_frac": frac_ex.get("all"), "executed_frac_conv2": frac_ex.get("conv2"),
                                 "aggregate_tflops": round(frac.get("all", 0.0) * PEAK_FP32_MFMA_TFLOPS, 2), "images_per_launch": 2 * B,
                                 "timing": "same instrumented pass as roofline (HIP events on the launch stream)"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(list(args.losses), full=args.cpu_baseline_full)
        print(json.dumps(out))
    if world > 1:
        _optim.destroy_native_comm()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
