"""What keeps a first run on eight GPUs from being a leap of faith, checked without eight GPUs (round 6; SURVEY.md 8e — the
reference is a single process on a single device, models/learner.py:65,187, so every line of this is the data-parallel wrapper's own):

  * `srlz.optim.rank_devices()` — under backend "nccl" two ranks on one device, or a process group whose size is not WORLD_SIZE, stop
    EVERY rank with a message (gloo, world 2; the backend name is the only thing stubbed);
  * `srlz.optim.all_ranks()` — a rank-local yes/no becomes the job's decision (the resident-store decision);
  * `srlz.optim.loader_workers()` — decoding threads x loader processes x local ranks fit the usable cores;
  * `srlz.optim.numa_cpus_of_device()` + `DataLoader(cpu_affinity=...)` — the producer process pins itself to the cores next to its GPU;
  * the per-epoch minibatch order of a multi-rank loader does not depend on how many occlusion draws a rank made.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(os.path.dirname(HERE), "srl-zoo_amd"), os.path.dirname(HERE), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_worker(rank, world, port, out_dir):
    for p in (os.path.join(os.path.dirname(HERE), "srl-zoo_amd"), os.path.dirname(HERE), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from srlz import optim
    res = {}
    # gloo: ranks may share a device (the debug topology) — the table is reported, nothing raises
    info = optim.rank_devices()
    res["gloo"] = (info["ranks"], info["devices"], info["hosts"])
    # the same job seen as an RCCL job: both ranks sit on one "device" -> every rank raises, with the shared device named
    real_backend = dist.get_backend
    dist.get_backend = lambda *a, **k: "nccl"
    identity = optim.device_identity
    try:
        try:
            optim.rank_devices()
            res["shared"] = "no error"
        except RuntimeError as e:
            res["shared"] = str(e)
        # one distinct device per rank: passes
        optim.device_identity = lambda index=None: ("host", "0000:%02x:00.0" % (0x10 + rank))
        ok = optim.rank_devices()
        res["distinct"] = (ok["ranks"], ok["devices"])
        # the launcher's WORLD_SIZE disagrees with the process group
        os.environ["WORLD_SIZE"] = str(world + 1)
        try:
            optim.rank_devices()
            res["world_mismatch"] = "no error"
        except RuntimeError as e:
            res["world_mismatch"] = str(e)
        os.environ["WORLD_SIZE"] = str(world)
    finally:
        dist.get_backend = real_backend
        optim.device_identity = identity
    # a rank-local verdict becomes the job's: only rank 1 says no -> nobody builds the store
    res["all_yes"] = optim.all_ranks(True)
    res["one_no"] = optim.all_ranks(rank != 1)
    torch.save(res, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_devices_and_collective_decisions(tmp_path):
    world = 2
    mp.spawn(_rank_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(str(tmp_path / ("rank%d.pt" % r)))
        assert res["gloo"] == (2, 1, 1)
        assert "2 ranks" in res["shared"] and "1 distinct GPU" in res["shared"] and "one GPU per rank" in res["shared"]
        assert res["distinct"] == (2, 2)
        assert "WORLD_SIZE=3" in res["world_mismatch"]
        assert res["all_yes"] is True and res["one_no"] is False


def test_loader_workers_fit_the_usable_cores():
    from srlz.optim import loader_workers
    # the reference's setting when the host is big enough: 8 ranks x 2 loader processes x 4 threads + 8 training threads <= 256 cores
    assert loader_workers(4, passes=2, local_ranks=8, cores=256) == 4
    # a 32-core host, eight ranks: (32 - 8) / 8 = 3 cores of loaders per rank -> 1 thread per process while the fill pass runs, 3 after
    assert loader_workers(4, passes=2, local_ranks=8, cores=32) == 1
    assert loader_workers(4, passes=1, local_ranks=8, cores=32) == 3
    assert loader_workers(4, passes=2, local_ranks=8, cores=96) == 4
    assert loader_workers(4, passes=2, local_ranks=8, cores=64) == 3
    assert loader_workers(4, passes=1, local_ranks=1, cores=2) == 1  # never below one
    for cores in (8, 16, 64, 256):
        for ranks in (1, 2, 4, 8):
            w = loader_workers(4, passes=2, local_ranks=ranks, cores=cores)
            assert 1 <= w <= 4 and (w == 1 or w * 2 * ranks + ranks <= cores)


def _fake_sysfs(root, pci, node, cpulist):
    d = root / "bus" / "pci" / "devices" / pci
    d.mkdir(parents=True)
    (d / "numa_node").write_text("%d\n" % node)
    if node >= 0:
        n = root / "devices" / "system" / "node" / ("node%d" % node)
        n.mkdir(parents=True)
        (n / "cpulist").write_text(cpulist + "\n")


def test_numa_cpus_of_device_reads_sysfs(tmp_path):
    from srlz.optim import numa_cpus_of_device
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip("needs two usable CPUs")
    half = allowed[:len(allowed) // 2]
    _fake_sysfs(tmp_path / "a", "0000:05:00.0", 1, "%d-%d,100000" % (half[0], half[-1]))
    got = numa_cpus_of_device("0000:05:00.0", sysfs=str(tmp_path / "a"))
    assert got == [c for c in allowed if half[0] <= c <= half[-1]]
    # no NUMA information / a node that covers everything this process may use / an unknown device: leave the mask alone
    _fake_sysfs(tmp_path / "b", "0000:05:00.0", -1, "")
    assert numa_cpus_of_device("0000:05:00.0", sysfs=str(tmp_path / "b")) is None
    _fake_sysfs(tmp_path / "c", "0000:05:00.0", 0, "0-%d" % (max(allowed) + 5))
    assert numa_cpus_of_device("0000:05:00.0", sysfs=str(tmp_path / "c")) is None
    assert numa_cpus_of_device("0000:77:00.0", sysfs=str(tmp_path / "c")) is None


def test_loader_process_pins_itself(tmp_path):
    """DataLoader(cpu_affinity=[c]) -> the forked producer (and the decoding threads it creates) runs on c only; the trainer's own
    mask is untouched."""
    from dataset_util import make_dataset
    from preprocessing.data_loader import DataLoader
    import preprocessing.preprocess as pre
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip("needs two usable CPUs")
    pre.N_CHANNELS = 3
    _, paths, _, _, _ = make_dataset(str(tmp_path), name="pin", n_episodes=1, ep_len=6)
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        loader = DataLoader([np.arange(0, 3)], paths, n_workers=2, is_training=False, infinite_loop=False, cpu_affinity=[allowed[1]])
        batch = next(loader)
        assert batch.shape[0] == 3
        assert os.sched_getaffinity(loader.process.pid) == {allowed[1]}
        assert sorted(os.sched_getaffinity(0)) == allowed
        loader.shutdown()
        assert loader.process is None
    finally:
        os.chdir(cwd)


def test_multi_rank_epoch_order_ignores_a_ranks_own_draws():
    """Two ranks fork their producers from the same RNG state; one of them makes a data-dependent number of other draws (the DAE's
    occlusion rectangles of ITS shard) between two epochs.  Their per-epoch permutations must stay the same permutation — the shards
    a partition of the epoch — which they are because the order comes from a private copy of the forked state."""
    from preprocessing.data_loader import DataLoader
    orders = []
    for rank, extra_draws in ((0, 0), (1, 37)):
        np.random.seed(1234)  # the state both producers are forked with
        loader = DataLoader.__new__(DataLoader)
        loader.shuffle, loader.n_minibatches, loader.rank, loader.world_size, loader.val_indices = True, 24, rank, 2, set([3, 9])
        loader._order_rng = np.random.RandomState()
        loader._order_rng.set_state(np.random.get_state())  # what _run does in the child
        epochs = []
        for _ in range(3):
            epochs.append(loader._epochOrder())
            for _ in range(extra_draws):
                np.random.randint(224)  # this rank's occlusion draws, from the global state
        orders.append(epochs)
    for e0, e1 in zip(*orders):
        both = np.concatenate((e0, e1))
        assert len(e0) == len(e1) and len(set(both.tolist())) == len(both)  # disjoint shards of one permutation
    # one rank: the reference's stream, the global state (np.random.permutation)
    np.random.seed(7)
    loader = DataLoader.__new__(DataLoader)
    loader.shuffle, loader.n_minibatches, loader.rank, loader.world_size, loader.val_indices, loader._order_rng = True, 10, 0, 1, None, None
    got = loader._epochOrder()
    np.random.seed(7)
    assert np.array_equal(got, np.random.permutation(10))
