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

Above the HBM budget (SRLZ_RESIDENT_GB, default: a third of the free device memory) the store lives in pinned host memory and a
minibatch is gathered by the host and copied — still no decode after epoch 1.
"""
from __future__ import print_function, division, absolute_import

import os

import numpy as np
import torch as th


def budget_bytes(device):
    env = os.environ.get("SRLZ_RESIDENT_GB")
    if env is not None:
        return int(float(env) * (1 << 30))
    free, _total = th.cuda.mem_get_info(device)
    return free // 3


class ResidentFrames(object):
    """uint8 frames [n_frames, C, W, H] kept across epochs, addressed by observation index."""

    def __init__(self, n_frames, frame_shape, device, needed, budget=None):
        """
        :param n_frames: (int) observations of the dataset (len(images_path))
        :param frame_shape: (C, W, H) of one planar uint8 frame
        :param device: (th.device) the GPU the training step runs on
        :param needed: (np.ndarray of int) observation indices the minibatches can ask for (idx and idx + 1 of every minibatch)
        :param budget: (int) bytes the store may take in HBM (None: budget_bytes(device))
        """
        from srlz import _cabi as C
        self.C = C
        self.device = device
        self.frame_shape = tuple(int(v) for v in frame_shape)
        self.frame_bytes = int(np.prod(self.frame_shape))
        if self.frame_bytes % 16:
            raise ValueError("frames of %d bytes: the index copies move 16-byte words" % self.frame_bytes)
        total = int(n_frames) * self.frame_bytes
        self.on_device = total <= (budget_bytes(device) if budget is None else budget)
        if self.on_device:
            self.store = th.empty((n_frames,) + self.frame_shape, dtype=th.uint8, device=device)
        else:
            self.store = th.empty((n_frames,) + self.frame_shape, dtype=th.uint8).pin_memory()
        self.have = np.zeros(n_frames, dtype=bool)
        self.needed = np.unique(np.asarray(needed, dtype=np.int64))
        self.missing = len(self.needed)
        self._need_mask = np.zeros(n_frames, dtype=bool)
        self._need_mask[self.needed] = True
        self.gathers = 0  # minibatches served from the store (tests / reports)

    def complete(self):
        return self.missing == 0

    def _index(self, idx):
        return th.from_numpy(np.ascontiguousarray(idx, dtype=np.int64)).to(self.device, non_blocking=True)

    def absorb(self, idx, obs, next_obs):
        """Keep the freshly decoded frames of minibatch `idx` (obs = frames idx, next_obs = frames idx + 1; uint8 [B, C, W, H] on the
        device).  Returns True once every needed frame is present."""
        from srlz.ops import stream, ptr
        idx = np.asarray(idx, dtype=np.int64)
        new0, new1 = ~self.have[idx], ~self.have[idx + 1]
        if new0.any() or new1.any():
            if self.on_device:
                di = self._index(idx)
                n = len(idx)
                for frames, shift in ((obs, 0), (next_obs, 1)):
                    frames = frames if frames.is_contiguous() else frames.contiguous()
                    self.C.copy_frames_u8(ptr(frames), None, 0, ptr(self.store), ptr(di), shift, n, self.frame_bytes, stream())
            else:
                host0, host1 = obs.cpu(), next_obs.cpu()
                self.store[th.from_numpy(idx)] = host0
                self.store[th.from_numpy(idx + 1)] = host1
            for sel in (idx[new0], idx[new1] + 1):
                fresh = sel[~self.have[sel]]
                self.have[fresh] = True
                self.missing -= int(self._need_mask[fresh].sum())
        return self.complete()

    def pair(self, idx):
        """(obs, next_obs) = (frames[idx], frames[idx + 1]) as the two halves of ONE device buffer (what the batched model call
        and the fused reconstruction loss want, cf. BaseLearner._toDevicePair)."""
        from srlz.ops import stream, ptr
        idx = np.asarray(idx, dtype=np.int64)
        n = len(idx)
        both = th.empty((2 * n,) + self.frame_shape, dtype=th.uint8, device=self.device)
        self.gathers += 1
        if self.on_device:
            di = self._index(idx)
            self.C.copy_frames_u8(ptr(self.store), ptr(di), 0, ptr(both[:n]), None, 0, n, self.frame_bytes, stream())
            self.C.copy_frames_u8(ptr(self.store), ptr(di), 1, ptr(both[n:]), None, 0, n, self.frame_bytes, stream())
        else:
            hi = th.from_numpy(idx)
            stage = th.empty((2 * n,) + self.frame_shape, dtype=th.uint8).pin_memory()
            th.index_select(self.store, 0, hi, out=stage[:n])
            th.index_select(self.store, 0, hi + 1, out=stage[n:])
            both.copy_(stage, non_blocking=True)
            both._srlz_stage = stage  # (keeps the pinned staging buffer alive until the copy has been consumed)
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
