"""End-to-end input rate of `train.py` on an on-disk dataset (GPU box): python tools/train_e2e.py [--frames 2000] [--epochs 5] [-bs 32 256]

Generates the BASELINE.md section 3 substitute of BASELINE.json configs[0]'s dataset (the reference does not vendor kuka_gym_test):
JPEG frames 224x224 in the reference's on-disk format, episodes of 250, 6 actions, seed 0 — then runs the product's unmodified
command line `python train.py --data-folder ... --losses autoencoder --model-type custom_cnn --state-dim 200 -bs B --epochs E`
and reports, per epoch, wall seconds and images/s from <log_folder>/epoch_stats.json:
  epoch 1  decodes every JPEG in the loader process (the reference's only mode: it does so in EVERY epoch),
  epoch 2+ gathers the resident frames by index (preprocessing/resident.py) and should sit at the bench.py rate.
`--no-resident` runs the same command with learner.RESIDENT_FRAMES = False (re-decoding every epoch, the reference's behaviour).
`--world W` starts W ranks of the same command line the way torch.distributed.run would (RCCL with one GPU per rank; on a box with
fewer GPUs the gloo debug topology, SRLZ_DIST_BACKEND=gloo, where the ranks share the GPU — each rank then has 1/W of it): every rank
decodes its slice of the dataset beside epoch 1, the slices are exchanged at the epoch boundary, and from epoch 2 on EVERY rank gathers
by index; the per-rank records come from <log_folder>/epoch_stats_rank<r>.json.
Prints one JSON document."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "tests")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2000)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("-bs", "--batch-sizes", type=int, nargs="+", default=[32, 256])
    ap.add_argument("--no-resident", action="store_true")
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--keep", default="")
    args = ap.parse_args()
    from dataset_util import make_dataset
    root = args.keep or tempfile.mkdtemp(prefix="srlz_e2e_")
    ep_len = 250
    t0 = time.time()
    if not os.path.exists(os.path.join(root, "data", "e2e", "ground_truth.npz")):
        make_dataset(root, name="e2e", n_episodes=max(1, args.frames // ep_len), ep_len=ep_len, seed=0)
    gen_s = time.time() - t0
    out = {"dataset": {"frames": (args.frames // ep_len) * ep_len, "episode_length": ep_len, "format": "JPEG 224x224, quality 95",
                       "generation_s": round(gen_s, 1)}, "resident": not args.no_resident, "runs": []}
    for bs in args.batch_sizes:
        log = os.path.join(root, "logs", "e2e_bs%d_%d" % (bs, int(not args.no_resident)))
        code = ("import sys, runpy; sys.path.insert(0, %r); import models.learner as L; L.RESIDENT_FRAMES = %r; "
                "sys.argv = ['train.py'] + sys.argv[1:]; runpy.run_path(%r, run_name='__main__')"
                % (os.path.join(REPO, "srl-zoo_amd"), not args.no_resident, os.path.join(REPO, "srl-zoo_amd", "train.py")))
        cmd = [sys.executable, "-c", code, "--no-display-plots", "--data-folder", "e2e", "--epochs", str(args.epochs), "--seed", "0",
               "--state-dim", "200", "--model-type", "custom_cnn", "-bs", str(bs), "--losses", "autoencoder", "--log-folder", log]
        t0 = time.time()
        if args.world == 1:
            proc = subprocess.run(cmd, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            returncode, text = proc.returncode, proc.stdout
        else:
            import socket
            import torch
            sock = socket.socket()
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
            sock.close()
            backend = "nccl" if torch.cuda.device_count() >= args.world else "gloo"
            procs = []
            for r in range(args.world):
                env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.world), LOCAL_WORLD_SIZE=str(args.world),
                           MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", SRLZ_DIST_BACKEND=backend)
                procs.append(subprocess.Popen(cmd, cwd=root, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                              stderr=subprocess.STDOUT))
            text = procs[0].communicate()[0]
            codes = [q.wait() for q in procs]
            returncode = max(codes, key=abs)
            out["world"] = {"ranks": args.world, "backend": backend, "gpus": torch.cuda.device_count()}
        wall = time.time() - t0
        rec = {"batch_size": bs, "returncode": returncode, "wall_s": round(wall, 1)}
        try:
            stats = json.load(open(os.path.join(log, "epoch_stats.json")))
            rec["epochs"] = [{"epoch": e["epoch"], "seconds": round(e["seconds"], 4), "images": e["images"],
                              "images_per_s": round(e["images"] / e["seconds"], 1), "index_minibatches": e["index_minibatches"],
                              "minibatches": e["minibatches"]} for e in stats]
            later = [e for e in rec["epochs"] if e["index_minibatches"] == e["minibatches"]]
            if later:
                rec["resident_epochs_images_per_s"] = round(sum(e["images"] for e in later) / sum(e["seconds"] for e in later), 1)
            if args.world > 1:
                rec["ranks"] = []
                for r in range(args.world):
                    st = json.load(open(os.path.join(log, "epoch_stats_rank%d.json" % r)))
                    ep = [{"epoch": e["epoch"], "seconds": round(e["seconds"], 4), "images": e["images"],
                           "images_per_s": round(e["images"] / e["seconds"], 1), "index_minibatches": e["index_minibatches"],
                           "minibatches": e["minibatches"]} for e in st]
                    lat = [e for e in ep if e["index_minibatches"] == e["minibatches"] and e["epoch"] >= 3]
                    rec["ranks"].append({"rank": r, "epochs": ep, "fill_wait_seconds": st[0].get("fill_wait_seconds"),
                                         "exchange": st[0].get("exchange"),
                                         "resident_epochs_images_per_s": round(sum(e["images"] for e in lat) /
                                                                               sum(e["seconds"] for e in lat), 1) if lat else None})
        except (IOError, OSError, ValueError) as e:
            rec["error"] = "%s\n%s" % (e, text.decode("utf-8", "replace")[-2000:])
        out["runs"].append(rec)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
