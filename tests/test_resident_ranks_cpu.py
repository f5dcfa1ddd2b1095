"""The resident dataset with several ranks (the reference's per-GPU stream is preprocessing/data_loader.py:129-193
inside the epoch loop of models/learner.py:354-372), driven on CPU with torch.distributed `gloo`: the product's scheduling objects —
DataLoader(rank, world_size, index_switch), ResidentFrames(rank, world_size), FillPass — around a STUB training step.

What must hold, at world_size 8 and 2:
  * the ranks' fill slices partition the dataset; a rank decodes only its slice beside epoch 1;
  * after the exchange at the FIRST epoch boundary every rank holds every frame, bit-identical to a fresh decode;
  * every rank switches to indices at that boundary: epoch 2 is index-only from its first minibatch, on every rank;
  * all ranks run the same number of steps per epoch (train first, then validation: the collectives stay in lock-step), and a
    rank's minibatches per epoch are disjoint from the other ranks';
  * when one rank's slice is incomplete NO rank switches (the decision is collective).
"""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dataset_util import make_dataset


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_fill_slices_partition_the_dataset():
    from preprocessing.resident import fill_slice, fill_minibatches
    for n in (1, 7, 8, 9, 104, 1000, 100001):
        for w in (1, 2, 3, 8):
            cover = []
            for r in range(w):
                lo, hi = fill_slice(n, r, w)
                assert 0 <= lo <= hi <= n
                chunks = fill_minibatches(n, r, w, chunk=64)
                flat = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.int64)
                assert flat.tolist() == list(range(lo, hi)) and all(0 < len(c) <= 64 for c in chunks)
                cover.extend(range(lo, hi))
            assert cover == list(range(n))
            sizes = [fill_slice(n, r, w)[1] - fill_slice(n, r, w)[0] for r in range(w)]
            assert max(sizes) == -(-n // w)  # nobody decodes more than ceil(n / W) frames


def _minibatches(n_frames, episode_starts, batch_size, val_size):
    """learn()'s minibatch construction (models/learner.py of this build = reference learner.py:253-283), same seed on every rank."""
    np.random.seed(0)
    indices = np.array([i for i in range(n_frames - 1) if not episode_starts[i + 1]], dtype='int64')
    np.random.shuffle(indices)
    mbl = [np.array(sorted(indices[s:s + batch_size])) for s in range(0, len(indices) - batch_size + 1, batch_size)]
    n_val = np.round(val_size * len(mbl)).astype(np.int64)
    val = np.random.permutation(len(mbl))[:n_val]
    return mbl, val


def _worker(rank, world, port, root, out_dir, sabotage_rank):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.join(os.path.dirname(here), "srl-zoo_amd"), os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.chdir(root)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from preprocessing.data_loader import DataLoader
    from preprocessing.resident import ResidentFrames, FillPass

    gt = np.load(os.path.join(root, "data", "ranks", "ground_truth.npz"))
    starts = np.load(os.path.join(root, "data", "ranks", "preprocessed_data.npz"))["episode_starts"]
    paths = gt["images_path"]
    n = len(paths)
    mbl, val = _minibatches(n, starts, batch_size=2, val_size=0.2)
    needed = np.concatenate([np.concatenate((mb, mb + 1)) for mb in mbl])
    resident = ResidentFrames(n, (3, 224, 224), "cpu", needed, rank=rank, world_size=world)
    assert not resident.on_device and resident.slice() == (min(rank * -(-n // world), n), min((rank + 1) * -(-n // world), n))
    fill = FillPass(resident, paths, n_workers=2, chunk=5)
    loader = DataLoader(mbl, paths, n_workers=2, is_training=True, rank=rank, world_size=world, val_indices=val,
                        raw_uint8="planar", index_switch=True, max_queue_len=2)
    val_set = set(int(v) for v in val)
    record = {"rank": rank, "epochs": []}

    def epoch():
        items = []
        for item in loader:  # (a stub step: nothing is computed; the collectives of a real step are the all-reduce below)
            kind = torch.tensor([1.0 if int(item[0]) in val_set else 0.0])
            dist.all_reduce(kind)  # every rank must be in the same phase (train / validation) at every step
            assert float(kind) in (0.0, float(world))
            if fill.loader is not None:
                fill.drain()
            items.append(item)
        return items

    # ---- epoch 1: pixels, as in the reference; the own slice arrives on the side
    first = epoch()
    assert all(it[1] is not None and it[1].dtype == torch.uint8 for it in first)
    lo, hi = resident.slice()
    if sabotage_rank == rank:
        fill.drain(block=True)
        resident.have[lo] = False  # "a frame of this rank's slice never arrived"
    switched = fill.finish(loader)
    record["switched"] = bool(switched)
    if sabotage_rank is not None:
        assert not switched and not loader.index_mode.is_set() and loader.epoch_gate.is_set()
        second = epoch()
        assert all(it[1] is not None for it in second) and len(second) == len(first)
    else:
        assert switched and resident.complete() and resident.have.all() and loader.index_mode.is_set()
        assert fill.stats["exchange"]["bytes"] == n * 3 * 224 * 224 and fill.stats["exchange"]["backend"] == "gloo"
        # every frame is home on every rank, bit-identical to a fresh decode (spot-check: first / last of every slice + own slice)
        probe = sorted(set([0, n - 1] + [resident.slice(r)[0] for r in range(world) if resident.slice(r)[0] < n] + list(range(lo, hi))))
        for i in probe:
            assert torch.equal(resident.store[i], DataLoader._makeBatchElement(paths[i], raw_uint8="planar")[0]), i
        digest = torch.tensor([int(resident.store.to(torch.int64).sum())])
        gathered = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(gathered, digest)
        assert all(torch.equal(gathered[0], g) for g in gathered)
        # ---- epochs 2 and 3: indices only, from the first minibatch on; the pair comes out of the store
        for _ in range(2):
            items = epoch()
            assert len(items) == len(first)
            assert all(it[1] is None and it[2] is None for it in items), [it[1] is None for it in items]
            mb = mbl[int(items[0][0])]
            obs, next_obs = resident.pair(mb)
            assert torch.equal(obs[1], resident.store[mb[1]]) and torch.equal(next_obs[0], resident.store[mb[0] + 1])
            record["epochs"].append([int(it[0]) for it in items])
        assert resident.gathers == 2
    record["first"] = [int(it[0]) for it in first]
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump(record, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.fixture(scope="module")
def ranks_dataset(tmp_path_factory):
    root = tmp_path_factory.mktemp("ranks")
    make_dataset(str(root), name="ranks", n_episodes=8, ep_len=14)
    return str(root)


@pytest.mark.parametrize("world", [8, 2])
def test_store_is_complete_on_every_rank_after_epoch_one(ranks_dataset, tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), ranks_dataset, str(tmp_path), None), nprocs=world, join=True)
    recs = [json.load(open(str(tmp_path / ("rank%d.json" % r)))) for r in range(world)]
    assert all(r["switched"] for r in recs)
    steps = {len(r["first"]) for r in recs} | {len(e) for r in recs for e in r["epochs"]}
    assert len(steps) == 1 and steps.pop() >= 2  # equal step counts per rank and epoch
    for e in range(2):
        seen = [i for r in recs for i in r["epochs"][e]]
        assert len(seen) == len(set(seen))  # the ranks' shards of an epoch are disjoint
    assert recs[0]["epochs"][0] != recs[0]["epochs"][1]  # a fresh permutation per epoch, from the forked RNG


def test_no_rank_switches_when_one_slice_is_incomplete(ranks_dataset, tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), ranks_dataset, str(tmp_path), 1), nprocs=2, join=True)
    recs = [json.load(open(str(tmp_path / ("rank%d.json" % r)))) for r in range(2)]
    assert [r["switched"] for r in recs] == [False, False]
