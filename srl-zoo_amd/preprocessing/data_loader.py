"""Per-GPU minibatch stream (reference preprocessing/data_loader.py:38-65, 68-280).

Same contract as the reference's DataLoader: a daemon producer process decodes and normalises images with a small
thread pool and pushes minibatches through a bounded queue; the training iterator yields
``(minibatch_idx, obs, next_obs, noisy_obs, next_noisy_obs)`` and ends each epoch with a ``None`` sentinel
(-> StopIteration); the test iterator yields single observation tensors.  Tensors are float32 CPU, reference layout
[B, C, W, H] (``transpose(0, 3, 2, 1)`` of an (H, W, C) image, reference :255).

Additions for data parallelism: ``rank`` / ``world_size`` — every rank draws the SAME per-epoch
permutation (same seed, same RNG state at fork) and consumes entries ``rank::world_size`` of it; the ragged tail is
dropped so all ranks run the same number of steps.

Image decoding is host-side I/O, not a kernel: OpenCV is used when importable (as in the reference), otherwise Pillow
(identical RGB pixels for the 224x224 JPEGs the datasets ship; INTER_AREA is approximated by a box filter otherwise).
"""
from __future__ import print_function, division, absolute_import

import gc
import glob
import os
import random
import time
from concurrent.futures import ThreadPoolExecutor

try:
    import queue
except ImportError:  # python 2
    import Queue as queue

import numpy as np
import torch as th
import torch.multiprocessing as _tmp

# The producer is a FORKED child by contract: it inherits the parent's np.random / random state (every rank forks from the same state
# and therefore draws the same per-epoch permutations, of which it takes its shard) and the loader object itself.  Whatever start
# method the host program has made the default (a process started by multiprocessing "spawn" inherits "spawn"), this module forks.
_mp = _tmp.get_context("fork")
Queue, Process, Event = _mp.Queue, _mp.Process, _mp.Event

from .preprocess import IMAGE_WIDTH, IMAGE_HEIGHT
from .utils import preprocessInput

try:
    import cv2
    _HAVE_CV2 = True
except ImportError:
    cv2 = None
    _HAVE_CV2 = False


def _imread_rgb(path):
    """Decode an image file to an RGB uint8 (H, W, 3) array resized to the network input; None if unreadable."""
    if _HAVE_CV2:
        im = cv2.imread(path)
        if im is None:
            return None
        im = cv2.resize(im, (IMAGE_WIDTH, IMAGE_HEIGHT), interpolation=cv2.INTER_AREA)
        return cv2.cvtColor(im, cv2.COLOR_BGR2RGB)
    from PIL import Image
    try:
        with Image.open(path) as f:
            im = f.convert("RGB")
            if im.size != (IMAGE_WIDTH, IMAGE_HEIGHT):
                im = im.resize((IMAGE_WIDTH, IMAGE_HEIGHT), Image.BOX)
            return np.asarray(im)
    except (IOError, OSError):
        return None


def sample_coordinates(coord_1, max_distance, percentage):
    """Second coordinate within [coord_1 -/+ max_distance*percentage], clipped to the axis (reference :23-35)."""
    low = max(0, coord_1 - max_distance * percentage)
    high = min(coord_1 + max_distance * percentage, max_distance)
    coord_2 = np.random.randint(low=low, high=high)
    return min(coord_1, coord_2), max(coord_1, coord_2)


def preprocessImage(image, convert_to_rgb=True, apply_occlusion=False, occlusion_percentage=0.5):
    """(H, W, 3) image -> normalised float32 (224, 224, 3); optional random rectangular occlusion (DAE).

    With OpenCV the input is BGR (cv2.imread) and is resized / converted here exactly as the reference does; without
    OpenCV the caller passes RGB and `convert_to_rgb` only applies to BGR inputs.
    """
    if _HAVE_CV2:
        im = cv2.resize(image, (IMAGE_WIDTH, IMAGE_HEIGHT), interpolation=cv2.INTER_AREA)
        if convert_to_rgb:
            im = cv2.cvtColor(im, cv2.COLOR_BGR2RGB)
    else:
        im = np.asarray(image)
        if convert_to_rgb:
            im = im[..., ::-1]
        if im.shape[:2] != (IMAGE_HEIGHT, IMAGE_WIDTH):
            from PIL import Image
            im = np.asarray(Image.fromarray(np.ascontiguousarray(im)).resize((IMAGE_WIDTH, IMAGE_HEIGHT), Image.BOX))
    im = preprocessInput(np.array(im, dtype=np.float32), mode="image_net")
    if apply_occlusion:
        h_1, h_2, w_1, w_2 = drawOcclusion(occlusion_percentage)
        im[h_1:h_2, w_1:w_2, :] = 0.
    return im


def drawOcclusion(occlusion_percentage):
    """The DAE's random rectangle (h_1, h_2, w_1, w_2) of one frame: the four np.random draws of the reference
    (preprocessing/data_loader.py:103-107), in its order."""
    h_1 = np.random.randint(IMAGE_HEIGHT)
    h_1, h_2 = sample_coordinates(h_1, IMAGE_HEIGHT, percentage=occlusion_percentage)
    w_1 = np.random.randint(IMAGE_WIDTH)
    w_1, w_2 = sample_coordinates(w_1, IMAGE_WIDTH, percentage=occlusion_percentage)
    return h_1, h_2, w_1, w_2


def _normalised(rgb, apply_occlusion, occlusion_percentage):
    # rgb: uint8 RGB already at network size
    return preprocessImage(rgb, convert_to_rgb=False, apply_occlusion=apply_occlusion,
                           occlusion_percentage=occlusion_percentage)


def shardOrder(order, rank, world_size, val_indices=None):
    """This rank's slice of a per-epoch minibatch order.  world_size 1: the order itself.  Otherwise training and
    validation minibatches are sharded separately (entries rank::world_size, ragged tails dropped, train first) so
    every rank runs the same number of optimisation steps and the gradient all-reduce never waits on a rank that is
    validating."""
    order = np.asarray(order, dtype=np.int64)
    if world_size == 1:
        return order
    val = val_indices if val_indices is not None else set()
    parts = []
    for keep_val in (False, True):
        sel = np.array([i for i in order if (int(i) in val) == keep_val], dtype=np.int64)
        usable = (len(sel) // world_size) * world_size
        parts.append(sel[:usable][rank::world_size])
    return np.concatenate(parts)


class DataLoader(object):
    def __init__(self, minibatchlist, images_path, n_workers=1, multi_view=False, use_triplets=False,
                 infinite_loop=True, max_queue_len=4, is_training=False, apply_occlusion=False,
                 occlusion_percentage=0.5, rank=0, world_size=1, val_indices=None, raw_uint8=False, index_switch=False,
                 cpu_affinity=None):
        """
        :param minibatchlist: ([np.array]) observation indices grouped per minibatch
        :param images_path: (np.array) image paths (without the 'data/' prefix)
        :param n_workers: (int) decoding threads
        :param multi_view: (bool) stack the two camera views along channels
        :param use_triplets: (bool) with multi_view: add a NEGATIVE observation (view 1 of a random other time step of the
                             same record) as channels 6..8 — the time-contrastive triplets of EmbeddingNet
        :param infinite_loop: (bool) restart after each epoch
        :param max_queue_len: (int) minibatches prepared ahead
        :param is_training: (bool) True: yield (idx, obs, next_obs, noisy, next_noisy) in shuffled order;
                            False: yield obs tensors in order
        :param apply_occlusion: (bool) also produce occluded copies (DAE)
        :param occlusion_percentage: (float)
        :param rank, world_size: data-parallel shard of the per-epoch permutation
        :param raw_uint8: yield the decoded frames as uint8 instead of normalised float32 [B, C, W, H], so that only a
                          quarter of the bytes cross PCIe.  True: [B, H, W, C] as decoded — the learner normalises and
                          transposes on the GPU (srlz_normalize_u8, bit-identical).  "planar": [B, C, W, H], i.e. the
                          reference's transpose(0, 3, 2, 1) (data_loader.py:255) applied by the worker to the uint8 frame —
                          the first convolution and the reconstruction loss then read the bytes themselves (srlz_conv1_fwd_u8
                          ...), no normalisation pass.  With occlusion (DAE) only "planar" is available: the clean frames
                          travel as bytes, the occluded copies as normalised float32.
        :param index_switch: (bool, training loaders) create the switch shipIndices() flips: from then on the producer puts
                          (minibatch_idx, None, None, rects, next_rects) — the SAME per-epoch permutation from the same forked RNG
                          and the same end-of-epoch None, but no pixels: the consumer holds the decoded frames
                          (preprocessing/resident.py) and gathers the minibatch by index.  rects / next_rects: with occlusion, the
                          int32 [B, views, 4] rectangles (h_1, h_2, w_1, w_2) drawn with the reference's np.random calls; with triplets,
                          the int64 [B] frame indices of the negative observations (the reference's random.randint draws); else None.
                          Such a producer also PAUSES behind the end-of-epoch marker of its first epoch until the consumer has said
                          what epoch 2 is made of — shipIndices() or keepPixels(); coming back for the next item without a decision
                          counts as keepPixels() — so the epoch after the decision is index-only from its first minibatch (the
                          producer runs max_queue_len + 1 minibatches ahead otherwise).  With several ranks the decision falls at
                          that boundary on every rank at once (preprocessing/resident.py::ResidentFrames.exchange).
        :param val_indices: minibatch ids used for validation; with world_size > 1 training and validation
                            minibatches are sharded separately (train first) so all ranks stay in lock-step
        :param cpu_affinity: (list of int or None) CPUs the producer process and its decoding threads pin themselves to — the cores of
                            the NUMA node the rank's GPU hangs off (srlz.optim.numa_cpus_of_device); None leaves the mask alone
        """
        super(DataLoader, self).__init__()
        if use_triplets and not multi_view:
            raise ValueError("triplets need the two camera views of a multi-view dataset")
        self.use_triplets = use_triplets
        self.n_workers = n_workers
        self.infinite_loop = infinite_loop
        self.n_minibatches = len(minibatchlist)
        self.minibatchlist = minibatchlist
        self.images_path = images_path
        self.shuffle = is_training
        self.queue = Queue(max_queue_len)
        self._max_queue_len, self._received, self._restarts = max_queue_len, 0, 0
        self.process = None
        self.multi_view = multi_view
        self.apply_occlusion = apply_occlusion
        self.occlusion_percentage = occlusion_percentage
        self.rank, self.world_size = rank, world_size
        self.cpu_affinity = None if not cpu_affinity else sorted(int(c) for c in cpu_affinity)
        self.val_indices = None if val_indices is None else set(int(i) for i in val_indices)
        if raw_uint8 and raw_uint8 != "planar" and apply_occlusion:
            raise ValueError("raw_uint8 frames cannot carry the (normalised-space) occlusion of the DAE loader")
        self.raw_uint8 = raw_uint8
        # (created BEFORE the fork: the producer sees the same flag)
        self._order_rng = None  # (several ranks: the producer's private copy of the forked RNG state, see _epochOrder)
        self.index_mode = Event() if (index_switch and is_training) else None
        self.epoch_gate = Event() if self.index_mode is not None else None
        self._index_of = None  # image stem -> frame index (built on first use by the triplet index path)
        self._epochs_received = 0
        self.startProcess()

    def shipIndices(self):
        """From the next minibatch the producer prepares: indices (and occlusion rectangles) instead of pixels."""
        if self.index_mode is None:
            raise ValueError("DataLoader(index_switch=True, is_training=True) creates the switch")
        self.index_mode.set()
        self.epoch_gate.set()

    def keepPixels(self):
        """The consumer's other answer at the first epoch boundary: keep decoding (the producer waits there for one of the two)."""
        if self.epoch_gate is not None:
            self.epoch_gate.set()

    @staticmethod
    def createTestMinibatchList(n_samples, batch_size):
        """Index ranges of at most batch_size covering n_samples (the last one may be empty, as in the reference)."""
        return [np.arange(i * batch_size, min(n_samples, (i + 1) * batch_size))
                for i in range(n_samples // batch_size + 1)]

    def stepsPerEpoch(self):
        """Minibatches this rank yields per epoch."""
        return len(shardOrder(np.arange(self.n_minibatches), self.rank, self.world_size, self.val_indices)) \
            if self.shuffle else self.n_minibatches

    def startProcess(self):
        self.process = Process(target=self._run)
        self.process.daemon = True  # dies with the trainer
        self.process.start()

    def _epochOrder(self):
        if not self.shuffle:
            return np.arange(self.n_minibatches, dtype=np.int64)
        # One rank: the reference's stream — np.random.permutation from the process's global state, interleaved with whatever else the
        # producer draws from it (occlusion rectangles).  Several ranks: every rank must draw the SAME permutation every epoch, but the
        # occlusion draws of a rank depend on its shard (np.random.randint rejects and redraws: a data-dependent number of variates),
        # so the order comes from a private copy of the state the producers were forked with — identical on all ranks, and touched by
        # nothing but these permutations.
        rng = np.random if self._order_rng is None else self._order_rng
        order = rng.permutation(self.n_minibatches).astype(np.int64)
        return shardOrder(order, self.rank, self.world_size, self.val_indices)

    def _run(self):
        # forked child: the parent's intra-op (OpenMP) worker threads do not exist here, so torch must not try to use
        # them — the few tensor ops below (tensor(), cat) run single-threaded; decoding parallelism comes from the pool
        # forked child of a process that owns a GPU: everything alive at the fork — including cyclic garbage of earlier training runs
        # that holds device tensors, streams and events — must stay untouched here.  Without the freeze this process's garbage
        # collector finalises such objects (hipFree / hipEventDestroy in a child that has no GPU context) and dies of a segmentation
        # fault while decoding, whenever a collection happens to fall into its first minibatch (found with faulthandler on a GPU box:
        # "Current thread: Garbage-collecting"; it showed as a training run waiting for ever or, since the watchdog, as exit code -11).
        gc.freeze()
        th.set_num_threads(1)
        if self.cpu_affinity and hasattr(os, "sched_setaffinity"):
            try:  # (threads created from here on — the decoding pool — inherit the mask)
                os.sched_setaffinity(0, self.cpu_affinity)
            except OSError:
                pass  # a mask the container does not allow: stay where we are
        if self.world_size > 1 and self.shuffle:
            self._order_rng = np.random.RandomState()
            self._order_rng.set_state(np.random.get_state())
        # ... and the decoding threads must not run ATen kernels at all: the OpenMP thread count is a per-thread setting, a pool thread
        # starts with the default (all cores), and a tensor copy above ATen's grain size opens a parallel region in this forked child,
        # whose inherited OpenMP runtime has no threads — it hangs or crashes (seen on GPU boxes with the DAE loaders).  The workers
        # therefore hand back numpy-made arrays wrapped by th.from_numpy (see _makeBatchElement); th.cat below runs on this thread.
        # (Calling th.set_num_threads from the workers is no cure: it rebuilds a process-wide thread pool, and four threads doing that
        # at once crashed the producer every time.)
        pool = ThreadPoolExecutor(max_workers=max(1, self.n_workers))
        first = True
        while first or self.infinite_loop:
            first = False
            for minibatch_idx in self._epochOrder():
                idx = self.minibatchlist[minibatch_idx]
                if self.index_mode is not None and self.index_mode.is_set():
                    rects = next_rects = None
                    if self.use_triplets:  # the negatives of the frames of obs, then of next_obs: indices into images_path
                        rects, next_rects = self._negativeIndices(idx), self._negativeIndices(idx + 1)
                    if self.apply_occlusion:  # one rectangle per frame and camera view, frames of obs first, then of next_obs
                        views = 2 if self.multi_view else 1
                        draw = np.array([drawOcclusion(self.occlusion_percentage) for _ in range(2 * len(idx) * views)],
                                        dtype=np.int32).reshape(2, len(idx), views, 4)
                        rects, next_rects = draw[0], draw[1]
                    self.queue.put((minibatch_idx, None, None, rects, next_rects))
                    continue
                if self.shuffle:
                    paths = np.concatenate((self.images_path[idx], self.images_path[idx + 1]))
                else:
                    paths = self.images_path[idx]
                clean = list(pool.map(lambda p: self._makeBatchElement(p, self.multi_view, self.use_triplets,
                                                                       raw_uint8=self.raw_uint8), paths))
                batch = th.cat(clean, dim=0) if clean else th.zeros(0)
                noisy = None
                if self.apply_occlusion:
                    occl = list(pool.map(lambda p: self._makeBatchElement(
                        p, self.multi_view, self.use_triplets, apply_occlusion=True,
                        occlusion_percentage=self.occlusion_percentage), paths))
                    noisy = th.cat(occl, dim=0)
                if self.shuffle:
                    half = len(paths) // 2
                    item = (minibatch_idx, batch[:half], batch[half:],
                            None if noisy is None else noisy[:half], None if noisy is None else noisy[half:])
                else:
                    item = batch
                self.queue.put(item)
            self.queue.put(None)  # end-of-epoch sentinel
            if self.epoch_gate is not None:
                self.epoch_gate.wait()  # (first epoch only: the event stays set) pixels or indices from here on?
        # one-shot loader: stay alive until terminated — tensors travel through the queue as shared-memory handles and
        # the sender must outlive their reception
        while True:
            time.sleep(0.05)

    # ---- time-contrastive triplets (reference data_loader.py:219-243): the negative observation of a frame is camera 1 of a random
    # OTHER time step of the same record.  The draw — a glob of the record's camera-1 files, the current step removed, one
    # random.randint — is kept in one place: the streaming path decodes the file it names, the index path (the dataset resident in
    # HBM) ships the INDEX of that time step in images_path instead.
    _record_steps = {}  # record prefix -> time steps in the order glob returned them (a fork inherits what the parent has seen)

    @classmethod
    def _negativeStep(cls, stem):
        """The time step (int) of the negative observation of 'data/<...>/frameNNNNNN' — the reference's draw, in its order."""
        extra_chars = '_1.jpg'
        prefix = stem[:-6]
        steps = cls._record_steps.get(prefix)
        if steps is None:
            digits_path = glob.glob(prefix + '[0-9]*' + extra_chars)
            steps = [int(k[:-len(extra_chars)][-6:]) for k in digits_path]
            cls._record_steps[prefix] = steps
        all_frame_steps = list(steps)
        all_frame_steps.remove(int(stem[-6:]))
        return all_frame_steps[random.randint(0, len(all_frame_steps) - 1)]

    @staticmethod
    def _stem(image_path):
        return 'data/' + image_path.split('.jpg')[0]

    @classmethod
    def negativesIndexable(cls, images_path):
        """True when every negative the reference could draw is itself an entry of images_path (every camera-1 file of every record is
        a listed time step): then a triplet is (index, negative index) into a store of the listed frames, and nothing else."""
        listed = {}
        for p in images_path:
            stem = cls._stem(p)
            listed.setdefault(stem[:-6], set()).add(int(stem[-6:]))
        for prefix, steps in listed.items():
            on_disk = set(int(k[:-len('_1.jpg')][-6:]) for k in glob.glob(prefix + '[0-9]*' + '_1.jpg'))
            if not on_disk or not on_disk <= steps:
                return False
        return True

    def _negativeIndices(self, idx):
        """Frame index (into images_path) of the negative observation of every frame in idx, drawn in order."""
        if self._index_of is None:
            self._index_of = {self._stem(p): i for i, p in enumerate(self.images_path)}
        out = np.empty(len(idx), dtype=np.int64)
        for k, i in enumerate(idx):
            stem = self._stem(self.images_path[i])
            out[k] = self._index_of['{}{:06d}'.format(stem[:-6], self._negativeStep(stem))]
        return out

    @classmethod
    def _makeBatchElement(cls, image_path, multi_view=False, use_triplets=False, apply_occlusion=False,
                          occlusion_percentage=None, raw_uint8=False):
        """One image path (without 'data/' prefix, '.jpg' optional) -> float32 tensor [1, C, W, H]
        (raw_uint8: the decoded RGB frame(s) as uint8 [1, H, W, C]; "planar": [1, C, W, H])."""
        stem = 'data/' + image_path.split('.jpg')[0]
        names = ["{}_{}.jpg".format(stem, i + 1) for i in range(2)] if multi_view else ["{}.jpg".format(stem)]
        occlude = [apply_occlusion] * len(names)
        if multi_view and use_triplets:
            # negative observation (reference data_loader.py:219-243): camera 1 of a random OTHER time step of the same
            # record, drawn with Python's `random` as the reference does; it is never occluded
            names.append('{}{:06d}_1.jpg'.format(stem[:-6], cls._negativeStep(stem)))
            occlude.append(False)
        views = []
        for name, occ in zip(names, occlude):
            rgb = _imread_rgb(name)
            if rgb is None:
                raise ValueError("tried to load {}, but it was not found".format(name))
            views.append(np.ascontiguousarray(rgb) if raw_uint8 else _normalised(rgb, occ, occlusion_percentage))
        im = np.dstack(views) if multi_view else views[0]
        if raw_uint8 == "planar":
            return th.from_numpy(np.ascontiguousarray(im.transpose(2, 1, 0)).reshape((1, im.shape[2], im.shape[1], im.shape[0])))
        if raw_uint8:
            return th.from_numpy(np.ascontiguousarray(im).reshape((1,) + im.shape))
        # channel first + batch dim; note the (W, H) order of the last two axes.  (The copy is numpy's: th.tensor() of a strided view
        # is an ATen copy kernel — an OpenMP parallel region in a pool thread of the forked producer, see _run.)
        return th.from_numpy(np.ascontiguousarray(im.reshape((1,) + im.shape).transpose(0, 3, 2, 1)))

    def __len__(self):
        return self.n_minibatches

    def __iter__(self):
        return self

    # A producer that DIES before its first item is re-forked (one that merely hangs: only with STARTUP_TIMEOUT > 0): fork() of a
    # multi-threaded parent (HIP runtime, OpenMP pools) can leave the child behind a lock some other thread owned at that instant
    # (the one crash that was actually found had another cause, see gc.freeze() in _run; this is the safety net).  The parent has drawn no random number
    # since the first fork, so the new child starts from the same RNG state and produces the same permutations.
    # A producer that dies AFTER it has delivered is reported instead of waited for: its epoch cannot be resumed.
    # Seconds a live producer may take for its first item before it is re-forked; 0 (the default) = never: only a DEAD producer is
    # replaced — a first minibatch can legitimately take minutes (cold network storage, bs = 256 pairs, the DAE decoding twice).
    STARTUP_TIMEOUT = float(os.environ.get("SRLZ_LOADER_STARTUP_TIMEOUT", "0"))
    MAX_RESTARTS = 3
    EMPTY = object()  # tryNext(): nothing ready yet

    def _restart(self, why):
        if self._restarts >= self.MAX_RESTARTS:
            raise RuntimeError("DataLoader: the producer process {} ({} attempts)".format(why, self._restarts + 1))
        self._restarts += 1
        try:
            self.process.terminate()
            self.process.join(5)
        except Exception:
            pass
        try:  # the abandoned queue's feeder thread and pipe
            self.queue.close()
            self.queue.cancel_join_thread()
        except Exception:
            pass
        self.queue = Queue(self._max_queue_len)
        self.startProcess()

    def _book(self, val):
        self._received += 1
        if val is None:
            self._epochs_received += 1
            raise StopIteration
        return val

    def _deadProducer(self):
        """The producer is gone.  One last look into the queue (it may have put its item and exited between our two checks), then:
        re-fork if nothing was ever delivered, report otherwise.  Returns an item or EMPTY (after a restart)."""
        try:
            return self.queue.get(timeout=0.05)
        except queue.Empty:
            pass
        if self._received == 0:
            self._restart("exited (code {}) before its first minibatch".format(self.process.exitcode))
            return self.EMPTY
        raise RuntimeError("DataLoader: the producer process exited (code {}) without finishing the epoch".format(
            self.process.exitcode))

    def _openGateIfUndecided(self):
        # the consumer is back for epoch 2 and has not said shipIndices() / keepPixels(): it wants what it had
        if self.epoch_gate is not None and self._epochs_received >= 1 and not self.epoch_gate.is_set():
            self.epoch_gate.set()

    def __next__(self):
        self._openGateIfUndecided()
        waited_since = time.time()
        while True:
            try:
                val = self.queue.get_nowait()
                break
            except queue.Empty:
                time.sleep(0.001)
                if time.time() - waited_since < 0.25:
                    continue
                if self.process is not None and not self.process.is_alive():
                    val = self._deadProducer()
                    if val is not self.EMPTY:
                        break
                    waited_since = time.time()
                    continue
                if self._received == 0 and self.STARTUP_TIMEOUT > 0 and time.time() - waited_since > self.STARTUP_TIMEOUT:
                    self._restart("delivered nothing in {} s".format(self.STARTUP_TIMEOUT))
                    waited_since = time.time()
        return self._book(val)

    def tryNext(self):
        """next() without the wait: an item, DataLoader.EMPTY when none is ready, StopIteration at the end-of-epoch marker."""
        self._openGateIfUndecided()
        try:
            val = self.queue.get_nowait()
        except queue.Empty:
            if self.process is not None and not self.process.is_alive():
                val = self._deadProducer()
                if val is self.EMPTY:
                    return val
            else:
                return self.EMPTY
        return self._book(val)

    next = __next__

    def shutdown(self):
        """End the producer process for good: terminate, JOIN (no zombie, its shared-memory handles released) and close the queue with
        its feeder thread and pipe.  For loaders that live shorter than the run (the slice's fill pass, the decode of the frames no
        minibatch asked for); the training loader dies with the trainer (daemon process)."""
        process, self.process = self.process, None
        if process is not None:
            try:
                process.terminate()
                process.join(5)
            except Exception:
                pass
        try:
            self.queue.close()
            self.queue.cancel_join_thread()
        except Exception:
            pass

    def __del__(self):
        try:
            if self.process is not None:
                self.process.terminate()
        except Exception:  # interpreter shutdown: multiprocessing internals may already be gone
            pass
