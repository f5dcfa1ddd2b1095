"""The per-GPU minibatch stream contract (reference preprocessing/data_loader.py:68-280): tuple format, tensor layout,
normalisation, epoch sentinel, test-mode iteration, data-parallel sharding of the per-epoch order."""
import os

import numpy as np
import pytest
import torch

from dataset_util import make_dataset


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    root = tmp_path_factory.mktemp("ds")
    info = make_dataset(str(root), n_episodes=2, ep_len=10)
    cwd = os.getcwd()
    os.chdir(str(root))
    yield info
    os.chdir(cwd)


def expected_tensor(path):
    from PIL import Image
    im = np.asarray(Image.open("data/" + path + ".jpg").convert("RGB")).astype(np.float32) / 255.0
    im = (im - np.array([0.485, 0.456, 0.406], np.float32)) / np.array([0.229, 0.224, 0.225], np.float32)
    return im.transpose(2, 1, 0)  # (C, W, H): the reference's transpose(0, 3, 2, 1) without the batch axis


def test_training_stream_contract(dataset):
    from preprocessing.data_loader import DataLoader
    name, paths, actions, rewards, starts = dataset
    minibatchlist = [np.array([0, 1, 2, 3]), np.array([4, 5, 6, 7]), np.array([10, 11, 12, 13])]
    loader = DataLoader(minibatchlist, paths, n_workers=2, is_training=True, infinite_loop=True)
    for epoch in range(2):
        seen = []
        for item in loader:
            idx, obs, next_obs, noisy, next_noisy = item
            assert noisy is None and next_noisy is None
            assert obs.dtype == torch.float32 and tuple(obs.shape) == (4, 3, 224, 224) and tuple(next_obs.shape) == (4, 3, 224, 224)
            mb = minibatchlist[int(idx)]
            np.testing.assert_allclose(obs[1].numpy(), expected_tensor(paths[mb[1]]), rtol=0, atol=1e-6)
            np.testing.assert_allclose(next_obs[2].numpy(), expected_tensor(paths[mb[2] + 1]), rtol=0, atol=1e-6)
            seen.append(int(idx))
        assert sorted(seen) == [0, 1, 2]  # every minibatch exactly once per epoch, then the None sentinel
    lo, hi = float(obs.min()), float(obs.max())
    assert lo >= -2.1180 and hi <= 2.6401  # value range of ImageNet-normalised pixels
    del loader


def test_test_stream_and_minibatch_list(dataset):
    from preprocessing.data_loader import DataLoader
    name, paths, *_ = dataset
    ml = DataLoader.createTestMinibatchList(len(paths), 8)
    assert [len(m) for m in ml] == [8, 8, 4]
    assert [len(m) for m in DataLoader.createTestMinibatchList(16, 8)] == [8, 8, 0]  # trailing empty range, as the reference
    loader = DataLoader(ml, paths, n_workers=2, is_training=False, max_queue_len=1, infinite_loop=False)
    batches = list(loader)
    assert [b.shape[0] for b in batches] == [8, 8, 4]
    np.testing.assert_allclose(batches[1][3].numpy(), expected_tensor(paths[11]), atol=1e-6)


def test_missing_image_raises(dataset):
    from preprocessing.data_loader import DataLoader
    with pytest.raises(ValueError):
        DataLoader._makeBatchElement("tiny_test/record_000/frame999999")


def test_preprocess_image_and_denormalize():
    from preprocessing.data_loader import preprocessImage
    from preprocessing.utils import deNormalize, preprocessInput
    rgb = np.random.RandomState(0).randint(0, 256, (224, 224, 3)).astype(np.uint8)
    out = preprocessImage(rgb, convert_to_rgb=False)
    ref = (rgb.astype(np.float32) / 255.0 - np.array([0.485, 0.456, 0.406], np.float32)) / np.array([0.229, 0.224, 0.225], np.float32)
    np.testing.assert_allclose(out, ref, atol=1e-6)
    back = deNormalize(out.transpose(2, 1, 0).copy())  # (3, W, H) -> (H, W, 3) in [0, 1]
    np.testing.assert_allclose(back, rgb / 255.0, atol=1e-6)
    x = np.full((2, 2, 3), 255.0, np.float32)
    assert np.allclose(preprocessInput(x.copy(), mode="tf"), 1.0)
    np.random.seed(0)
    occl = preprocessImage(rgb, convert_to_rgb=False, apply_occlusion=True, occlusion_percentage=0.5)
    assert (occl == 0).all(axis=2).any()  # a zeroed rectangle exists


def test_shard_order_lockstep():
    from preprocessing.data_loader import shardOrder
    order = np.random.RandomState(1).permutation(23)
    val = {3, 7, 11, 19, 22}
    assert np.array_equal(shardOrder(order, 0, 1, val), order)
    shards = [shardOrder(order, r, 4, val) for r in range(4)]
    assert len({len(s) for s in shards}) == 1  # same number of steps on every rank
    flat = np.concatenate(shards)
    assert len(set(flat.tolist())) == len(flat)  # disjoint
    for step in range(len(shards[0])):
        kinds = {int(s[step]) in val for s in shards}
        assert len(kinds) == 1  # all ranks train, or all ranks validate, at every step
    n_train = sum(1 for i in shards[0] if int(i) not in val)
    assert all(int(i) not in val for i in shards[0][:n_train]) and all(int(i) in val for i in shards[0][n_train:])


def test_raw_uint8_stream(dataset):
    """raw_uint8=True yields the decoded frames themselves; normalising them on the host reproduces the float stream."""
    from preprocessing.data_loader import DataLoader
    from preprocessing.utils import preprocessInput
    name, paths, *_ = dataset
    ml = [np.array([0, 1, 2]), np.array([4, 5, 6])]
    raw = DataLoader(ml, paths, n_workers=2, is_training=True, raw_uint8=True)
    idx, obs, next_obs, noisy, next_noisy = next(iter(raw))
    assert obs.dtype == torch.uint8 and tuple(obs.shape) == (3, 224, 224, 3) and noisy is None
    ref = preprocessInput(obs[1].numpy().astype(np.float32), mode="image_net").transpose(2, 1, 0)
    np.testing.assert_array_equal(ref, expected_tensor(paths[ml[int(idx)][1]]))
    with pytest.raises(ValueError):
        DataLoader(ml, paths, is_training=True, raw_uint8=True, apply_occlusion=True)
    del raw
    # raw_uint8="planar": the same bytes in the reference's tensor layout (transpose(0, 3, 2, 1), data_loader.py:255) — what
    # learn() ships to the kernels that normalise while staging (srlz_conv1_fwd_u8 ...)
    for i in (0, 5):
        a = DataLoader._makeBatchElement(paths[i], raw_uint8=True)
        b = DataLoader._makeBatchElement(paths[i], raw_uint8="planar")
        assert b.dtype == torch.uint8 and tuple(b.shape) == (1, 3, 224, 224) and b.is_contiguous()
        assert torch.equal(b, a.permute(0, 3, 2, 1))
    planar = DataLoader(ml, paths, n_workers=2, is_training=True, raw_uint8="planar")
    idx, obs, next_obs, noisy, next_noisy = next(iter(planar))
    assert obs.dtype == torch.uint8 and tuple(obs.shape) == (3, 3, 224, 224) and tuple(next_obs.shape) == (3, 3, 224, 224)
    ref = preprocessInput(obs[1].numpy().transpose(2, 1, 0).astype(np.float32), mode="image_net").transpose(2, 1, 0)
    np.testing.assert_array_equal(ref, expected_tensor(paths[ml[int(idx)][1]]))
    del planar


def test_triplet_stream(tmp_path):
    """multi_view + use_triplets (reference data_loader.py:207-245): channels 0-2 = camera 1, 3-5 = camera 2 of the frame,
    6-8 = camera 1 of ANOTHER time step of the same record (the negative), in float and in raw-uint8 form."""
    from PIL import Image
    from preprocessing.data_loader import DataLoader
    name, paths, *_ = make_dataset(str(tmp_path), name="tiny_mv", n_episodes=2, ep_len=6, multi_view=True)
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        def view(path, v):
            im = np.asarray(Image.open("data/%s_%d.jpg" % (path, v)).convert("RGB")).astype(np.float32) / 255.0
            im = (im - np.array([0.485, 0.456, 0.406], np.float32)) / np.array([0.229, 0.224, 0.225], np.float32)
            return im.transpose(2, 1, 0)
        el = DataLoader._makeBatchElement(paths[2], multi_view=True, use_triplets=True)
        assert tuple(el.shape) == (1, 9, 224, 224)
        np.testing.assert_allclose(el[0, :3].numpy(), view(paths[2], 1), atol=1e-6)
        np.testing.assert_allclose(el[0, 3:6].numpy(), view(paths[2], 2), atol=1e-6)
        others = [p for p in paths[:6] if p != paths[2]]  # the other frames of record 0
        assert any(np.allclose(el[0, 6:].numpy(), view(p, 1), atol=1e-6) for p in others)
        assert not np.allclose(el[0, 6:].numpy(), view(paths[2], 1), atol=1e-6)
        import random
        random.seed(5)
        raw = DataLoader._makeBatchElement(paths[8], multi_view=True, use_triplets=True, raw_uint8=True)
        assert raw.dtype == torch.uint8 and tuple(raw.shape) == (1, 224, 224, 9)
        random.seed(5)  # (the negative view is drawn with Python's `random`: same draw -> same frame)
        planar = DataLoader._makeBatchElement(paths[8], multi_view=True, use_triplets=True, raw_uint8="planar")
        assert planar.dtype == torch.uint8 and tuple(planar.shape) == (1, 9, 224, 224) and torch.equal(planar, raw.permute(0, 3, 2, 1))
        loader = DataLoader([np.array([0, 1]), np.array([6, 7])], paths, n_workers=2, multi_view=True, use_triplets=True,
                            is_training=True, infinite_loop=False)
        items = list(loader)
        assert len(items) == 2 and all(tuple(it[1].shape) == (2, 9, 224, 224) and tuple(it[2].shape) == (2, 9, 224, 224) for it in items)
        with pytest.raises(ValueError):
            DataLoader([np.array([0])], paths, use_triplets=True)  # needs multi_view
    finally:
        os.chdir(cwd)


def test_index_mode_keeps_the_iteration_contract(dataset):
    """DataLoader(index_switch=True).shipIndices(): the SAME sequence of minibatch ids per epoch as a pixel-shipping loader forked
    with the same RNG state (np.random.permutation per epoch in the child), the same end-of-epoch marker, no pixels; with occlusion
    the items carry one int32 rectangle (h_1, h_2, w_1, w_2) per frame and view, inside the image and ordered."""
    from preprocessing.data_loader import DataLoader
    name, paths, *_ = dataset
    ml = [np.array([0, 1, 2]), np.array([4, 5, 6]), np.array([8, 9, 10]), np.array([12, 13, 14]), np.array([16, 17, 18])]

    def epochs(loader, count):
        out = []
        for _ in range(count):
            out.append([item for item in loader])  # (StopIteration at the None marker; the producer goes on with the next epoch)
        return out

    np.random.seed(11)
    pixels = DataLoader(ml, paths, n_workers=2, is_training=True, raw_uint8="planar", max_queue_len=2)
    np.random.seed(11)
    indices = DataLoader(ml, paths, n_workers=2, is_training=True, raw_uint8="planar", max_queue_len=2, index_switch=True)
    indices.shipIndices()
    a, b = epochs(pixels, 3), epochs(indices, 3)
    assert [[int(i[0]) for i in e] for e in a] == [[int(i[0]) for i in e] for e in b]
    assert all(sorted(int(i[0]) for i in e) == list(range(5)) for e in b)
    flat = [i for e in b for i in e]
    # (the producer may have prepared its first minibatches before the switch was flipped: pixels are legal, indices must follow)
    first_idx = next(k for k, i in enumerate(flat) if i[1] is None)
    assert first_idx <= 3 and all(i[1] is None and i[2] is None and i[3] is None and i[4] is None for i in flat[first_idx:])
    del pixels, indices
    with pytest.raises(ValueError):
        DataLoader(ml, paths, is_training=True).shipIndices()
    # DAE: clean frames as bytes + occluded copies as floats while streaming; rectangles once indices are shipped
    np.random.seed(5)
    dae = DataLoader(ml, paths, n_workers=2, is_training=True, raw_uint8="planar", apply_occlusion=True, occlusion_percentage=0.4,
                     max_queue_len=1, index_switch=True)
    idx, obs, next_obs, noisy, next_noisy = next(iter(dae))
    if obs is not None:
        assert obs.dtype == torch.uint8 and noisy.dtype == torch.float32 and tuple(noisy.shape) == (3, 3, 224, 224)
        assert (noisy == 0).any()
    dae.shipIndices()
    seen = 0
    for _ in range(12):
        try:
            item = next(dae)
        except StopIteration:
            continue
        if item[1] is None:
            seen += 1
            for r in (item[3], item[4]):
                assert r.dtype == np.int32 and r.shape == (3, 1, 4)
                assert (0 <= r[..., 0]).all() and (r[..., 0] <= r[..., 1]).all() and (r[..., 1] <= 224).all()
                assert (0 <= r[..., 2]).all() and (r[..., 2] <= r[..., 3]).all() and (r[..., 3] <= 224).all()
    assert seen >= 3


def test_producer_that_dies_before_its_first_minibatch_is_reforked(dataset, tmp_path):
    """fork() of a multi-threaded trainer occasionally yields a child that crashes at once (seen on GPU boxes with the DAE loader: exit
    code -11).  A producer that dies before delivering anything is re-forked — same RNG state at fork, hence the same permutation —
    and one that dies after having delivered raises instead of leaving the training loop waiting for ever."""
    import os
    import signal
    from preprocessing.data_loader import DataLoader
    name, paths, *_ = dataset
    ml = [np.array([0, 1, 2]), np.array([4, 5, 6]), np.array([8, 9, 10])]
    flag = str(tmp_path / "crash_once")
    open(flag, "w").close()

    class CrashOnce(DataLoader):
        def _run(self):
            if os.path.exists(flag):
                os.remove(flag)
                os.kill(os.getpid(), signal.SIGSEGV)
            return DataLoader._run(self)

    np.random.seed(21)
    good = DataLoader(ml, paths, n_workers=2, is_training=True, raw_uint8="planar")
    order = [int(i[0]) for i in good]
    del good
    np.random.seed(21)
    dl = CrashOnce(ml, paths, n_workers=2, is_training=True, raw_uint8="planar")
    assert [int(i[0]) for i in dl] == order and dl._restarts == 1 and not os.path.exists(flag)

    class DiesLater(DataLoader):
        def _run(self):
            self.queue.put((0, None, None, None, None))
            time.sleep(0.3)
            os.kill(os.getpid(), signal.SIGSEGV)

    import time
    late = DiesLater(ml, paths, is_training=True)
    assert next(late)[0] == 0
    with pytest.raises(RuntimeError, match="exited"):
        next(late)


def test_epoch_gate_makes_the_second_epoch_index_only_from_its_first_minibatch(dataset):
    """DataLoader(index_switch=True): the producer pauses behind its first end-of-epoch marker until the consumer has decided —
    shipIndices(): every minibatch of epoch 2 is an index item (no look-ahead pixels); keepPixels() or simply coming back for more:
    pixels as before.  The sequence of minibatch ids is the ungated loader's either way."""
    from preprocessing.data_loader import DataLoader
    name, paths, *_ = dataset
    ml = [np.array([0, 1, 2]), np.array([4, 5, 6]), np.array([8, 9, 10]), np.array([12, 13, 14]), np.array([16, 17, 18])]
    np.random.seed(3)
    plain = DataLoader(ml, paths, n_workers=2, is_training=True, raw_uint8="planar", max_queue_len=4)
    want = [[int(i[0]) for i in plain] for _ in range(3)]
    del plain

    np.random.seed(3)
    gated = DataLoader(ml, paths, n_workers=2, is_training=True, raw_uint8="planar", max_queue_len=4, index_switch=True)
    first = [i for i in gated]
    assert all(i[1] is not None for i in first)
    import time
    time.sleep(0.5)  # a producer running ahead would have queued epoch-2 pixels by now
    assert gated.queue.empty() and not gated.epoch_gate.is_set()
    gated.shipIndices()
    second, third = [i for i in gated], [i for i in gated]
    assert all(i[1] is None and i[2] is None for i in second + third)
    assert [[int(i[0]) for i in e] for e in (first, second, third)] == want
    del gated

    for decide in ("keep", "undecided"):
        np.random.seed(3)
        dl = DataLoader(ml, paths, n_workers=2, is_training=True, raw_uint8="planar", max_queue_len=4, index_switch=True)
        a = [i for i in dl]
        if decide == "keep":
            dl.keepPixels()
        b = [i for i in dl]  # (undecided: asking for the next item opens the gate)
        assert dl.epoch_gate.is_set() and all(i[1] is not None for i in a + b)
        assert [[int(i[0]) for i in e] for e in (a, b)] == want[:2]
        del dl


def test_try_next_never_waits(dataset):
    from preprocessing.data_loader import DataLoader
    name, paths, *_ = dataset
    ranges = [np.arange(0, 3), np.arange(3, 5), np.arange(5, 5)]
    dl = DataLoader(ranges, paths, n_workers=2, is_training=False, infinite_loop=False, raw_uint8="planar", max_queue_len=4)
    import time
    got, empties, t0 = [], 0, time.time()
    while time.time() - t0 < 60:
        try:
            item = dl.tryNext()
        except StopIteration:
            break
        if item is DataLoader.EMPTY:
            empties += 1
            time.sleep(0.005)
            continue
        got.append(item)
    assert [tuple(g.shape) for g in got] == [(3, 3, 224, 224), (2, 3, 224, 224), (0,)] and empties >= 1
    assert torch.equal(got[1][0], DataLoader._makeBatchElement(paths[3], raw_uint8="planar")[0])
    assert DataLoader.STARTUP_TIMEOUT == 0.0  # a live producer is never re-forked unless SRLZ_LOADER_STARTUP_TIMEOUT says so


def test_triplet_index_mode_names_the_negatives_the_streaming_loader_decodes(tmp_path):
    """The triplets of the resident store (round 5): with indices shipped, an item carries the FRAME INDEX of every negative observation
    — drawn by the loader process with the reference's glob + random.randint (data_loader.py:219-243).
    (a) the two code paths make the same draw: for the same state of Python's `random`, the element builder of the streaming path decodes
        camera 1 of exactly the frame the index path names;
    (b) through the loader processes (whose `random` is re-seeded by Python at fork — in the reference as here — so the draws of two
        processes cannot be compared): same minibatch order as a streaming loader forked from the same np.random state, negatives are
        other time steps of the frame's own record, obs and next_obs draws are separate."""
    import random
    from preprocessing.data_loader import DataLoader
    name, paths, *_ = make_dataset(str(tmp_path), name="tiny_tri", n_episodes=2, ep_len=7, multi_view=True)
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        assert DataLoader.negativesIndexable(paths)
        assert not DataLoader.negativesIndexable(paths[:5])  # a listed subset: negatives on disk that are not frames of the dataset
        ml = [np.array([0, 1, 2]), np.array([3, 4, 5]), np.array([7, 8, 9]), np.array([10, 11, 12])]
        np.random.seed(3)
        stream = DataLoader(ml, paths, n_workers=1, multi_view=True, use_triplets=True, is_training=True, raw_uint8="planar",
                            infinite_loop=False)
        np.random.seed(3)
        index = DataLoader(ml, paths, n_workers=1, multi_view=True, use_triplets=True, is_training=True, raw_uint8="planar",
                           infinite_loop=False, index_switch=True)
        index.shipIndices()
        # (a) in this process, on the loader object itself
        for seed in range(6):
            frames = np.array([1, 4, 9, 12])
            random.seed(seed)
            named = index._negativeIndices(frames)
            random.seed(seed)
            for k, f in enumerate(frames):
                el = DataLoader._makeBatchElement(paths[f], multi_view=True, use_triplets=True, raw_uint8="planar")
                view1 = DataLoader._makeBatchElement(paths[named[k]], multi_view=True, raw_uint8="planar")[0, :3]
                assert torch.equal(el[0, 6:], view1) and named[k] != f
        # (b) through the processes
        a, b = list(stream), list(index)
        assert [int(i[0]) for i in a] == [int(i[0]) for i in b] and len(a) == 4
        assert sum(1 for i in b if i[1] is None) >= 3  # (the producer may have prepared its first minibatch before the switch was flipped)
        assert all(tuple(i[1].shape) == (3, 9, 224, 224) for i in a)
        for i_item in b:
            if i_item[1] is not None:
                continue
            mb = ml[int(i_item[0])]
            neg, next_neg = i_item[3], i_item[4]
            assert neg.dtype == np.int64 and neg.shape == (3,) and next_neg.shape == (3,)
            for k in range(3):
                for idx, nn in ((mb[k], neg[k]), (mb[k] + 1, next_neg[k])):
                    assert nn != idx and paths[nn].rsplit("/", 1)[0] == paths[idx].rsplit("/", 1)[0]  # another step of the same record
    finally:
        os.chdir(cwd)
