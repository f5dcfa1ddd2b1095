"""The dataset resident in HBM.

The reference's loader process decodes every JPEG of every epoch again (preprocessing/data_loader.py:195-256 inside the epoch loop
of models/learner.py:354-372).  At the rate this build trains (tens of thousands of images per second and GPU) that loader is more
than an order of magnitude too slow — and unnecessary: 100 000 frames x 150 KB of planar uint8 are 15 GB, a twentieth of one
MI355X's HBM.  So every frame is decoded ONCE:

  * epoch 1 streams as before (the loader process decodes, frames cross PCIe as bytes); every minibatch that arrives is also
    scattered into the store (`absorb`: one srlz_copy_frames_u8 launch per frame set);
  * once every frame a training / validation minibatch can ask for is present, the loader process is told to ship INDICES only
    (`DataLoader.index_mode`): same permutation per epoch from the same forked RNG, same None sentinel — only the pixels stay
    home — and the step's pair buffer [obs ; next_obs] is gathered on the device by index (`pair`);
  * the DAE's occluded copies are made on the device from the resident frames and the rectangles the loader process drew
    (`occluded_pair`, srlz_occlude_frames_u8).

Several ranks (one process per GPU, replicated data parallelism: the reference's stream is per GPU, data_loader.py:129-193).  A rank's
epoch-1 stream carries only its shard of the minibatches (data_loader.py::shardOrder: a fresh random 1/W of them every epoch), so
waiting to have SEEN every frame would take ~ln(M)/ln(W/(W-1)) epochs.  Instead every rank owns a fixed contiguous slice of the
dataset (`fill_slice`): a second loader process decodes exactly that slice beside the first training epoch (`fill_minibatches` ->
`absorb_range`), and at the end of epoch 1 the slices are exchanged — `exchange`: the owner broadcasts its slice in chunks, in place
into every rank's store, over the process group the gradients use (RCCL over xGMI; a 15 GB dataset crosses in about a second).
Every rank then holds the whole dataset and flips its loader to indices at the SAME epoch boundary.  Decoding work of epoch 1 per
rank: its 2/W of the pair stream + 1/W for the slice, instead of the whole dataset.

Above the HBM budget (SRLZ_RESIDENT_GB, default: a third of the free device memory) the store lives in pinned host memory and a
minibatch is gathered by the host and copied — still no decode after epoch 1.
"""
from __future__ import print_function, division, absolute_import

import os
import time

import numpy as np
import torch as th


def budget_bytes(device):
    env = os.environ.get("SRLZ_RESIDENT_GB")
    if env is not None:
        return int(float(env) * (1 << 30))
    if th.device(device).type != "cuda":
        return 0
    free, _total = th.cuda.mem_get_info(device)
    return free // 3


def fill_slice(n_frames, rank, world_size):
    """[lo, hi): the frames rank `rank` decodes for everybody.  Contiguous, ceil(n / W) frames each, the last ones shorter (or empty)."""
    per = -(-int(n_frames) // int(world_size))
    lo = min(rank * per, n_frames)
    return lo, min(lo + per, n_frames)


def fill_minibatches(n_frames, rank, world_size, chunk=64):
    """The rank's slice as consecutive index ranges of at most `chunk` frames: the minibatch list of the fill loader
    (DataLoader(is_training=False, infinite_loop=False, raw_uint8="planar") yields one uint8 tensor per range, in this order)."""
    lo, hi = fill_slice(n_frames, rank, world_size)
    return [np.arange(a, min(a + chunk, hi), dtype=np.int64) for a in range(lo, hi, chunk)]


class FillPass(object):
    """A rank's side pass over its own slice of the dataset during the first training epoch (world_size > 1), and the decision at
    the epoch boundary.  Used by SRL4robotics.learn(); tests/test_resident_ranks_cpu.py drives the same object with a stub step.

        fill = FillPass(resident, images_path, n_workers, multi_view)     # forks the slice's loader process
        for minibatch in epoch 1: step(...); fill.drain()                 # absorb whatever chunks are decoded by now
        fill.finish(train_loader)    # COLLECTIVE: rest of the slice, exchange, then shipIndices() / keepPixels() on every rank alike
    """

    def __init__(self, resident, images_path, n_workers=4, multi_view=False, chunk=64, cpu_affinity=None):
        from .data_loader import DataLoader
        self.resident = resident
        self.ranges = resident.fillMinibatches(chunk)
        self._next_range = 0
        self.loader = DataLoader(self.ranges, images_path, n_workers=n_workers, multi_view=multi_view, is_training=False,
                                 infinite_loop=False, max_queue_len=4, raw_uint8="planar", cpu_affinity=cpu_affinity)
        self.done = False
        self.stats = {}

    def drain(self, block=False):
        """Move decoded slice chunks into the store (block: until the loader's end-of-slice marker).  True once the slice is in."""
        while not self.done:
            try:
                frames = next(self.loader) if block else self.loader.tryNext()
            except StopIteration:
                self.done = True
                break
            if frames is self.loader.EMPTY:
                break
            rng = self.ranges[self._next_range]  # (the loader yields one uint8 tensor per index range, in list order)
            self._next_range += 1
            self.resident.absorb_range(int(rng[0]) if len(rng) else 0, frames)
        return self.done

    def finish(self, train_loader):
        """The first epoch boundary — every rank gets here after the same number of steps: wait for the rest of the own slice,
        exchange the slices, and tell the training loader what epoch 2 is made of.  Returns True when the ranks switched to indices."""
        t0 = time.time()
        self.drain(block=True)
        self.stats["fill_wait_seconds"] = time.time() - t0
        switched = self.resident.exchange()
        if switched:
            train_loader.shipIndices()
            self.stats["exchange"] = self.resident.exchange_stats
        else:
            train_loader.keepPixels()
        self.close()
        return switched

    def close(self):
        loader, self.loader = self.loader, None
        if loader is not None:
            loader.shutdown()


class ResidentFrames(object):
    """uint8 frames [n_frames, C, W, H] kept across epochs, addressed by observation index."""

    EXCHANGE_CHUNK_BYTES = 256 << 20

    @staticmethod
    def fits_device(n_frames, frame_shape, device, budget=None):
        """Whether a store of n_frames x frame_shape bytes would live in HBM (else: pinned host memory)."""
        total = int(n_frames) * int(np.prod(frame_shape))
        return th.device(device).type == "cuda" and total <= (budget_bytes(device) if budget is None else budget)

    def __init__(self, n_frames, frame_shape, device, needed, budget=None, rank=0, world_size=1):
        """
        :param n_frames: (int) observations of the dataset (len(images_path))
        :param frame_shape: (C, W, H) of one planar uint8 frame
        :param device: (th.device) the GPU the training step runs on
        :param needed: (np.ndarray of int) observation indices the minibatches can ask for (idx and idx + 1 of every minibatch)
        :param budget: (int) bytes the store may take in HBM (None: budget_bytes(device))
        :param rank, world_size: the data-parallel job (world_size > 1: the store is completed by `exchange`, not by `absorb`)
        """
        self.device = th.device(device)
        self.rank, self.world_size = int(rank), int(world_size)
        self.n_frames = int(n_frames)
        self.frame_shape = tuple(int(v) for v in frame_shape)
        self.frame_bytes = int(np.prod(self.frame_shape))
        if self.frame_bytes % 16:
            raise ValueError("frames of %d bytes: the index copies move 16-byte words" % self.frame_bytes)
        self.on_device = self.fits_device(self.n_frames, self.frame_shape, self.device, budget)
        if self.on_device:
            from srlz import _cabi as C
            self.C = C
            self.store = th.empty((self.n_frames,) + self.frame_shape, dtype=th.uint8, device=self.device)
        else:
            # (pinned at allocation: .pin_memory() of a pageable tensor would hold the dataset twice for a moment)
            self.store = th.empty((self.n_frames,) + self.frame_shape, dtype=th.uint8, pin_memory=th.cuda.is_available())
        self.have = np.zeros(self.n_frames, dtype=bool)
        self.needed = np.unique(np.asarray(needed, dtype=np.int64))
        self.missing = len(self.needed)
        self._need_mask = np.zeros(self.n_frames, dtype=bool)
        self._need_mask[self.needed] = True
        self.gathers = 0  # minibatches served from the store (tests / reports)
        self.exchange_stats = None
        # host store: two pinned staging buffers for the gathered pair alternate, each guarded by the event of the copy that read it
        self._stage, self._stage_ev, self._stage_i = [None, None], [None, None], 0

    def complete(self):
        return self.missing == 0

    def _mark(self, fresh):
        fresh = fresh[~self.have[fresh]]
        self.have[fresh] = True
        self.missing -= int(self._need_mask[fresh].sum())

    def _index(self, idx):
        return th.from_numpy(np.ascontiguousarray(idx, dtype=np.int64)).to(self.device, non_blocking=True)

    def absorb(self, idx, obs, next_obs):
        """Keep the freshly decoded frames of minibatch `idx` (obs = frames idx, next_obs = frames idx + 1; uint8 [B, C, W, H] on the
        device).  Returns True once every needed frame is present."""
        idx = np.asarray(idx, dtype=np.int64)
        new0, new1 = ~self.have[idx], ~self.have[idx + 1]
        if new0.any() or new1.any():
            if self.on_device:
                from srlz.ops import stream, ptr
                di = self._index(idx)
                n = len(idx)
                for frames, shift in ((obs, 0), (next_obs, 1)):
                    frames = frames if frames.is_contiguous() else frames.contiguous()
                    got = int(frames[0].numel())  # (triplet minibatches carry a third view behind the two the store keeps)
                    if got == self.frame_bytes:
                        self.C.copy_frames_u8(ptr(frames), None, 0, ptr(self.store), ptr(di), shift, n, self.frame_bytes, stream())
                    else:
                        self.C.copy_frames_u8_strided(ptr(frames), None, 0, got, 0, ptr(self.store), ptr(di), shift, self.frame_bytes, 0,
                                                      n, self.frame_bytes, stream())
            else:
                c = self.frame_shape[0]
                host0, host1 = obs[:, :c].cpu(), next_obs[:, :c].cpu()
                self.store[th.from_numpy(idx)] = host0
                self.store[th.from_numpy(idx + 1)] = host1
            self._mark(idx[new0])
            self._mark(idx[new1] + 1)
        return self.complete()

    # ---- several ranks: own slice decoded beside epoch 1, slices exchanged at its end ------------------------------------------
    def slice(self, rank=None):
        return fill_slice(self.n_frames, self.rank if rank is None else rank, self.world_size)

    def fillMinibatches(self, chunk=64):
        return fill_minibatches(self.n_frames, self.rank, self.world_size, chunk)

    def absorb_range(self, start, frames):
        """Frames start .. start + len(frames) - 1 as the fill loader decoded them (uint8 [n, C, W, H], host or device)."""
        n = int(frames.shape[0])
        if n == 0:
            return
        if tuple(frames.shape[1:]) != self.frame_shape or frames.dtype != th.uint8 or start < 0 or start + n > self.n_frames:
            raise ValueError("absorb_range: %s frames %s at %d do not fit a store of %d x %s" % (
                frames.dtype, tuple(frames.shape), start, self.n_frames, self.frame_shape))
        self.store[start:start + n].copy_(frames, non_blocking=True)
        self._mark(np.arange(start, start + n, dtype=np.int64))

    def absorb_indices(self, idx, frames):
        """Frames of the observations idx (any order; uint8 [n, C, W, H], host or device) — e.g. the few a run's minibatches never
        asked for, decoded when the states of the whole dataset are predicted from the store."""
        idx = np.asarray(idx, dtype=np.int64)
        n = len(idx)
        if n == 0:
            return
        if tuple(frames.shape) != (n,) + self.frame_shape or frames.dtype != th.uint8:
            raise ValueError("absorb_indices: %s frames %s for %d indices of a %s store" % (frames.dtype, tuple(frames.shape), n,
                                                                                            self.frame_shape))
        if self.on_device:
            from srlz.ops import stream, ptr
            frames = frames.to(self.device).contiguous()
            self.C.copy_frames_u8(ptr(frames), None, 0, ptr(self.store), ptr(self._index(idx)), 0, n, self.frame_bytes, stream())
        else:
            self.store[th.from_numpy(idx)] = frames.cpu()
        self._mark(idx)

    def slice_filled(self):
        lo, hi = self.slice()
        return bool(self.have[lo:hi].all())

    def exchange(self):
        """COLLECTIVE (every rank, same point of the program): each rank's slice reaches every other rank's store.  Returns True when
        the store is complete on EVERY rank afterwards — the ranks must take the switch to indices together — and False (nothing
        moved) when some rank's fill did not finish.

        The owner broadcasts its slice chunk by chunk, in place: no gather buffer, no second copy of the dataset.  Process group nccl
        (RCCL): device pointers — the store itself, or a device staging chunk when the store lives in host memory.  gloo (the debug
        topology of srlz.optim.dist_backend, and the CPU tests): host pointers — the store itself, or a host bounce of the chunk."""
        if self.world_size == 1:
            return self.complete()
        import torch.distributed as dist
        t0 = time.time()
        device_coll = dist.get_backend() == "nccl"
        coll_dev = self.device if device_coll else th.device("cpu")
        ok = th.tensor([1 if self.slice_filled() else 0], dtype=th.int32, device=coll_dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) != 1:
            return False
        per = max(1, self.EXCHANGE_CHUNK_BYTES // self.frame_bytes)
        direct = (device_coll and self.on_device) or (not device_coll and not self.on_device)
        moved = 0
        stage = None
        for owner in range(self.world_size):
            lo, hi = self.slice(owner)
            for a in range(lo, hi, per):
                b = min(a + per, hi)
                view = self.store[a:b]
                if direct:
                    dist.broadcast(view, src=owner)
                else:
                    if stage is None:
                        stage = th.empty((per,) + self.frame_shape, dtype=th.uint8, device=coll_dev)
                    buf = stage[:b - a]
                    if owner == self.rank:
                        buf.copy_(view)
                    dist.broadcast(buf, src=owner)
                    if owner != self.rank:
                        view.copy_(buf)
                moved += (b - a) * self.frame_bytes
        if self.device.type == "cuda":
            th.cuda.synchronize(self.device)
        self.have[:] = True
        self.missing = 0
        self.exchange_stats = {"seconds": time.time() - t0, "bytes": moved, "chunks_of": per, "direct": bool(direct),
                               "backend": dist.get_backend()}
        return True

    # ---- the step's input ---------------------------------------------------------------------------------------------------
    def pair(self, idx):
        """(obs, next_obs) = (frames[idx], frames[idx + 1]) as the two halves of ONE device buffer (what the batched model call
        and the fused reconstruction loss want, cf. BaseLearner._toDevicePair)."""
        idx = np.asarray(idx, dtype=np.int64)
        n = len(idx)
        both = th.empty((2 * n,) + self.frame_shape, dtype=th.uint8, device=self.device)
        self.gathers += 1
        if self.on_device:
            from srlz.ops import stream, ptr
            di = self._index(idx)
            self.C.copy_frames_u8(ptr(self.store), ptr(di), 0, ptr(both[:n]), None, 0, n, self.frame_bytes, stream())
            self.C.copy_frames_u8(ptr(self.store), ptr(di), 1, ptr(both[n:]), None, 0, n, self.frame_bytes, stream())
        else:
            hi = th.from_numpy(idx)
            k = self._stage_i
            self._stage_i ^= 1
            if self._stage_ev[k] is not None:
                self._stage_ev[k].synchronize()  # the copy that last read this staging buffer (two steps ago) is done
            # (the staging buffers are flat byte tensors in pair() and triplet_pair() alike: a store may serve both)
            if self._stage[k] is None or self._stage[k].numel() < both.numel():
                self._stage[k] = th.empty(both.numel(), dtype=th.uint8, pin_memory=th.cuda.is_available())
            stage = self._stage[k][:both.numel()].view(both.shape)
            th.index_select(self.store, 0, hi, out=stage[:n])
            th.index_select(self.store, 0, hi + 1, out=stage[n:])
            both.copy_(stage, non_blocking=True)
            if self.device.type == "cuda":
                self._stage_ev[k] = th.cuda.Event()
                self._stage_ev[k].record(th.cuda.current_stream(self.device))
        return both[:n], both[n:]

    def triplet_pair(self, idx, negatives, next_negatives):
        """The time-contrastive triplet observations of a minibatch (reference preprocessing/data_loader.py:219-243) from a store of
        two-view frames [n_frames, 6, W, H]: (obs, next_obs), each [B, 9, W, H] = [view 1 ; view 2 ; view 1 of the negative time step],
        as the halves of one buffer.  negatives / next_negatives: int64 [B] frame indices drawn by the loader process."""
        idx = np.asarray(idx, dtype=np.int64)
        n = len(idx)
        c, w, h = self.frame_shape
        if c != 6:
            raise ValueError("triplet_pair reads a store of two-view frames (6 channels), not %d" % c)
        view = 3 * w * h
        both = th.empty((2 * n, 9, w, h), dtype=th.uint8, device=self.device)
        self.gathers += 1
        if self.on_device:
            from srlz.ops import stream, ptr
            di = self._index(idx)
            for half, (shift, neg) in enumerate(((0, negatives), (1, next_negatives))):
                out = both[half * n:(half + 1) * n]
                dn = self._index(neg)
                self.C.copy_frames_u8_strided(ptr(self.store), ptr(di), shift, 2 * view, 0, ptr(out), None, 0, 3 * view, 0, n, 2 * view,
                                              stream())
                self.C.copy_frames_u8_strided(ptr(self.store), ptr(dn), 0, 2 * view, 0, ptr(out), None, 0, 3 * view, 2 * view, n, view,
                                              stream())
        else:
            k = self._stage_i
            self._stage_i ^= 1
            if self._stage_ev[k] is not None:
                self._stage_ev[k].synchronize()
            if self._stage[k] is None or self._stage[k].numel() < both.numel():
                self._stage[k] = th.empty(both.numel(), dtype=th.uint8, pin_memory=th.cuda.is_available())
            stage = self._stage[k][:both.numel()].view(both.shape)
            for half, (shift, neg) in enumerate(((0, negatives), (1, next_negatives))):
                out = stage[half * n:(half + 1) * n]
                out[:, :6] = self.store[th.from_numpy(idx + shift)]
                out[:, 6:] = self.store[th.from_numpy(np.asarray(neg, dtype=np.int64))][:, :3]
            both.copy_(stage, non_blocking=True)
            if self.device.type == "cuda":
                self._stage_ev[k] = th.cuda.Event()
                self._stage_ev[k].record(th.cuda.current_stream(self.device))
        return both[:n], both[n:]

    def occluded_pair(self, idx, rects, next_rects):
        """The DAE's noisy (obs, next_obs): normalised frames with one random rectangle per camera view set to 0, as float32 halves of
        one buffer.  rects / next_rects: int arrays [B, views, 4] = (h1, h2, w1, w2) drawn by the loader process."""
        from srlz.ops import stream, ptr, norm_lut
        if not self.on_device:
            raise RuntimeError("occluded_pair needs the store in HBM")
        idx = np.asarray(idx, dtype=np.int64)
        n = len(idx)
        c, w, h = self.frame_shape
        di = self._index(idx)
        out = th.empty((2 * n, c, w, h), dtype=th.float32, device=self.device)
        lut = norm_lut(self.device)
        for half, (r, shift) in enumerate(((rects, 0), (next_rects, 1))):
            rd = th.from_numpy(np.ascontiguousarray(r, dtype=np.int32).reshape(n, c // 3, 4)).to(self.device, non_blocking=True)
            self.C.occlude_frames_u8(ptr(self.store), ptr(di), shift, ptr(rd), ptr(lut), ptr(out[half * n:(half + 1) * n]), n, c, w, h,
                                     stream())
        return out[:n], out[n:]
