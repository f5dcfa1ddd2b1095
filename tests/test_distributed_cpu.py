"""The N > 1 path on CPU: two processes, torch.distributed `gloo` (world_size 2) — one all-reduce of the flat gradient
bucket per step (gradients + the scalar tail: a NaN loss on one rank is seen by all), identical parameters on every rank
afterwards, lock-step sharding of the minibatch order, rank-averaged BatchNorm statistics in a checkpoint, one log folder."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.join(os.path.dirname(here), "srl-zoo_amd"), os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import preprocessing.preprocess as pre
    from models.modules import SRLModules
    from preprocessing.data_loader import shardOrder
    from srlz import optim
    pre.N_CHANNELS = 3
    np.random.seed(1)
    torch.manual_seed(1)  # same seed on every rank -> identical initial parameters
    model = SRLModules(state_dim=8, action_dim=4, model_type="custom_cnn", losses=["autoencoder"])
    flat = optim.FlatParams(model)
    assert optim.world() == (rank, world)
    # rank-specific "gradients": what each rank's backward would have left in the flat bucket
    g = torch.Generator().manual_seed(100 + rank)
    local = torch.randn(flat.grad.numel(), generator=g)
    flat.grad.copy_(local)
    scale = optim.allreduce_gradients(flat)  # ONE collective over the whole bucket
    assert scale == 1.0 / world
    expect = sum(torch.randn(flat.grad.numel(), generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
    assert torch.allclose(flat.grad, expect, rtol=0, atol=1e-6)
    # a plain-torch Adam step with the averaged gradient (the HIP Adam kernel applies `scale` itself on the GPU)
    m = torch.zeros_like(flat.flat)
    v = torch.zeros_like(flat.flat)
    gavg = flat.grad * scale
    m.mul_(0.9).add_(gavg, alpha=0.1)
    v.mul_(0.999).addcmul_(gavg, gavg, value=0.001)
    flat.flat.addcdiv_(m / (1 - 0.9), (v / (1 - 0.999)).sqrt().add_(1e-8), value=-1e-3)
    # parameters (views of the flat buffer) must be bit-identical across ranks
    digest = torch.tensor([float(flat.flat.double().sum()), float(flat.flat.double().abs().sum())], dtype=torch.float64)
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    assert all(torch.equal(gathered[0], t) for t in gathered)
    w = dict(model.named_parameters())["model.encoder_conv.4.weight"]
    assert w.data_ptr() >= flat.flat.data_ptr()  # still a view of the updated buffer
    # minibatch order: same permutation everywhere, disjoint lock-step shards
    order = np.random.RandomState(5).permutation(21)
    val = {2, 9, 13, 20}
    mine = shardOrder(order, rank, world, val)
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([len(mine)]))
    assert len({int(s) for s in sizes}) == 1
    np.save(os.path.join(out_dir, "shard%d.npy" % rank), mine)

    # ---- the scalar tail: every rank reads back the SAME mean losses; one rank's NaN reaches all (exit code 11 together)
    flat.zero_grad()
    flat.put_scalars([torch.tensor(2.0 + rank), torch.tensor(10.0 * (rank + 1))])
    optim.allreduce_gradients(flat)
    assert flat.read_scalars(2) == [2.5, 15.0]
    flat.put_scalars([torch.tensor(float("nan") if rank == 1 else 1.0), torch.tensor(1.0)])
    optim.allreduce_scalars(flat)  # (a validation minibatch: no gradients exchanged)
    vals = flat.read_scalars(2)
    assert np.isnan(vals[0]) and vals[1] == 1.0
    # ---- checkpoint: BatchNorm running statistics are the ranks' average, counters and parameters untouched
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    key = "model.encoder_conv.1.running_mean"
    sd[key] = torch.full_like(sd[key], float(rank))
    sd["model.encoder_conv.1.running_var"] = torch.full_like(sd[key], 1.0 + 2.0 * rank)
    avg = optim.average_running_stats(sd)
    assert torch.allclose(avg[key], torch.full_like(avg[key], 0.5))
    assert torch.allclose(avg["model.encoder_conv.1.running_var"], torch.full_like(avg[key], 2.0))
    assert avg["model.encoder_conv.1.num_batches_tracked"].dtype == torch.long
    assert torch.equal(avg["model.encoder_conv.0.weight"], model.state_dict()["model.encoder_conv.0.weight"])
    # ---- one log folder for all ranks (train.py): rank 0's choice
    assert optim.share_from_rank0("logs/run_%d" % rank) == "logs/run_0"
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_exchange_and_sharding(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    shards = [np.load(str(tmp_path / ("shard%d.npy" % r))) for r in range(world)]
    assert len(set(np.concatenate(shards).tolist())) == sum(len(s) for s in shards)
    val = {2, 9, 13, 20}
    for a, b in zip(*shards):
        assert (int(a) in val) == (int(b) in val)
