"""The dataset resident in HBM (GPU; SURVEY.md 8 f-1, srl-zoo_amd/preprocessing/resident.py): decode once, gather by index.

  * ResidentFrames: scatter of arriving minibatches, gather of [obs ; next_obs] as two halves of one buffer (device store and the
    pinned-host fallback), completion bookkeeping over exactly the frames the minibatches can ask for;
  * the DAE's occluded copies made on the device == the loader's host arithmetic (preprocessInput, then the rectangle set to 0);
  * SRL4robotics.learn() with the store == learn() re-decoding every epoch (the reference's behaviour), number for number: the
    loader process ships the same permutation as indices, the gathered bytes are the decoded bytes."""
import os

import numpy as np
import pytest
import torch

from dataset_util import make_dataset

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    root = tmp_path_factory.mktemp("resident")
    info = make_dataset(str(root), n_episodes=4, ep_len=26)
    cwd = os.getcwd()
    os.chdir(str(root))
    yield info
    os.chdir(cwd)


@pytest.mark.parametrize("budget", [None, 0], ids=["hbm", "pinned_host"])
def test_store_scatter_gather_round_trip(budget):
    from preprocessing.resident import ResidentFrames
    rs = np.random.RandomState(3)
    n_frames, shape = 40, (3, 224, 224)
    frames = torch.from_numpy(rs.randint(0, 256, (n_frames,) + shape).astype(np.uint8))
    ml = [np.array([0, 1, 2, 5]), np.array([8, 9, 30, 31]), np.array([2, 6, 7, 33])]
    needed = np.concatenate([np.concatenate((m, m + 1)) for m in ml])
    store = ResidentFrames(n_frames, shape, torch.device(DEV, torch.cuda.current_device()), needed, budget=budget)
    assert store.on_device == (budget is None) and not store.complete()
    done = []
    for m in ml:
        done.append(store.absorb(m, frames[m].to(DEV), frames[m + 1].to(DEV)))
    assert done == [False, False, True] and store.complete() and int(store.have.sum()) == len(np.unique(needed))
    assert store.absorb(ml[0], frames[ml[0]].to(DEV), frames[ml[0] + 1].to(DEV))  # absorbing again changes nothing
    for m in ml[::-1]:
        obs, nxt = store.pair(m)
        torch.cuda.synchronize()
        assert obs.dtype == torch.uint8 and obs.is_cuda and tuple(obs.shape) == (4,) + shape
        assert torch.equal(obs.cpu(), frames[m]) and torch.equal(nxt.cpu(), frames[m + 1])
        # the two halves of ONE buffer (what the batched model call and the fused reconstruction loss take without a copy)
        assert obs.untyped_storage().data_ptr() == nxt.untyped_storage().data_ptr()
        assert nxt.storage_offset() == obs.storage_offset() + obs.numel()
    assert store.gathers == 3


@pytest.mark.parametrize("budget", [None, 0], ids=["hbm", "pinned_host"])
def test_triplet_pair_from_a_two_view_store(budget):
    """Time-contrastive triplets (reference preprocessing/data_loader.py:219-243) from the resident store: the store keeps the two
    camera views of every time step (6 channels); a streamed 9-channel minibatch leaves its first two views there (strided scatter),
    and a triplet minibatch is [views of frame idx ; view 1 of the frame the loader named as negative] (two strided gathers)."""
    from preprocessing.resident import ResidentFrames
    rs = np.random.RandomState(11)
    n_frames = 24
    two = torch.from_numpy(rs.randint(0, 256, (n_frames, 6, 224, 224)).astype(np.uint8))
    ml = [np.array([0, 1, 2, 5]), np.array([8, 9, 20, 21]), np.array([12, 13, 14, 16])]
    needed = np.concatenate([np.concatenate((m, m + 1)) for m in ml])
    store = ResidentFrames(n_frames, (6, 224, 224), torch.device(DEV, torch.cuda.current_device()), needed, budget=budget)
    for m in ml:  # what the streaming epoch delivers: 9 channels, the third view a negative that is NOT kept
        junk = torch.from_numpy(rs.randint(0, 256, (len(m), 3, 224, 224)).astype(np.uint8))
        obs = torch.cat((two[m], junk), 1).to(DEV)
        nxt = torch.cat((two[m + 1], junk), 1).to(DEV)
        store.absorb(m, obs, nxt)
    assert store.complete()
    for m in ml:
        neg, next_neg = rs.choice(np.unique(needed), len(m)), rs.choice(np.unique(needed), len(m))
        obs, nxt = store.triplet_pair(m, neg, next_neg)
        torch.cuda.synchronize()
        assert tuple(obs.shape) == (4, 9, 224, 224) and obs.dtype == torch.uint8 and obs.is_cuda
        assert torch.equal(obs[:, :6].cpu(), two[m]) and torch.equal(obs[:, 6:].cpu(), two[neg][:, :3])
        assert torch.equal(nxt[:, :6].cpu(), two[m + 1]) and torch.equal(nxt[:, 6:].cpu(), two[next_neg][:, :3])
        assert nxt.storage_offset() == obs.storage_offset() + obs.numel()  # the halves of one buffer
    with pytest.raises(ValueError):
        ResidentFrames(4, (3, 224, 224), torch.device(DEV, torch.cuda.current_device()), np.arange(4)).triplet_pair(
            np.array([0]), np.array([1]), np.array([2]))


@pytest.mark.parametrize("c", [3, 6])
def test_device_occlusion_is_the_loaders(c):
    """srlz_occlude_frames_u8 on resident frames == preprocessInput(frame) with im[h_1:h_2, w_1:w_2, :] = 0 per camera view, in the
    reference's [C, W, H] tensor layout (preprocessing/data_loader.py:100-111,255), bit for bit."""
    from preprocessing.resident import ResidentFrames
    from preprocessing.utils import preprocessInput
    rs = np.random.RandomState(c)
    n_frames, views = 9, c // 3
    hwc = rs.randint(0, 256, (n_frames, 224, 224, c)).astype(np.uint8)            # frames as decoded
    planar = torch.from_numpy(np.ascontiguousarray(hwc.transpose(0, 3, 2, 1)))     # [N, C, W, H]
    idx = np.array([0, 3, 4, 6])
    store = ResidentFrames(n_frames, (c, 224, 224), torch.device(DEV, torch.cuda.current_device()), np.arange(n_frames))
    store.absorb(np.arange(0, 8), planar[0:8].to(DEV), planar[1:9].to(DEV))
    assert store.complete()

    def rects():
        r = np.zeros((len(idx), views, 4), dtype=np.int32)
        for i in range(len(idx)):
            for v in range(views):
                h = np.sort(rs.randint(0, 225, 2))
                w = np.sort(rs.randint(0, 225, 2))
                r[i, v] = (h[0], h[1], w[0], w[1])
        r[0, 0] = (0, 224, 0, 224)   # everything occluded
        r[1, 0] = (17, 17, 5, 200)   # empty rectangle
        return r
    r0, r1 = rects(), rects()
    noisy, next_noisy = store.occluded_pair(idx, r0, r1)
    torch.cuda.synchronize()
    for got, r, shift in ((noisy, r0, 0), (next_noisy, r1, 1)):
        for i, f in enumerate(idx + shift):
            ref = np.concatenate([preprocessInput(hwc[f, :, :, 3 * v:3 * v + 3].astype(np.float32), mode="image_net")
                                  for v in range(views)], axis=2)
            for v in range(views):
                h1, h2, w1, w2 = r[i, v]
                ref[h1:h2, w1:w2, 3 * v:3 * v + 3] = 0.
            np.testing.assert_array_equal(got[i].cpu().numpy(), ref.transpose(2, 1, 0))


@pytest.mark.parametrize("losses", [["autoencoder", "inverse"], ["dae"]], ids=["ae_inverse", "dae"])
def test_learn_with_resident_frames_is_learn_with_redecoding(workdir, losses):
    import models.learner as learner
    import preprocessing.preprocess as pre
    from models.learner import SRL4robotics
    name, paths, actions, rewards, starts = workdir
    pre.N_CHANNELS = 3
    learner.N_EPOCHS, learner.BATCH_SIZE, learner.VALIDATION_SIZE, learner.DISPLAY_PLOTS = 4, 8, 0.2, False

    def run(resident):
        learner.RESIDENT_FRAMES = resident
        try:
            log = "logs/res_%s_%d" % (losses[0], int(resident))
            os.makedirs(log, exist_ok=True)
            srl = SRL4robotics(10, model_type="custom_cnn", seed=4, learning_rate=1e-3, cuda=True, losses=losses, n_actions=6,
                               log_folder=log, occlusion_percentage=0.3)
            hist, states, _ = srl.learn(paths, actions, rewards, starts)
            # the final states: from the store (the frames no minibatch asked for decoded then) unless the model looks at occluded
            # frames (DAE) or the run re-decodes anyway
            assert srl.predict_stats["from_store"] == (resident and losses != ["dae"]), srl.predict_stats
            if srl.predict_stats["from_store"]:
                assert srl.predict_stats["decoded_now"] < 16 and srl._resident.have.all()
            return {k: list(v) for k, v in hist.items()}, states, srl._resident
        finally:
            learner.RESIDENT_FRAMES = True

    h1, s1, store = run(True)
    h0, s0, none = run(False)
    assert none is None and store is not None and store.complete() and store.on_device
    # 12 minibatches per epoch, 4 epochs: epoch 1 streams; the producer waits at the first epoch boundary for the decision, so epochs
    # 2-4 are index-only from their first minibatch (at most the tail of epoch 1 too, when the store completed before its last step)
    assert 36 <= store.gathers <= 40, store.gathers
    assert sorted(h1) == sorted(h0)
    if losses == ["dae"]:
        # the occlusion rectangles are random draws of the loader process, raced between its decoding threads in the reference
        # itself: no two runs see the same rectangles -> the runs agree in law only (the device-side occlusion itself is held bit
        # for bit in test_device_occlusion_is_the_loaders)
        for k in h0:
            assert np.isfinite(h1[k]).all() and len(h1[k]) == len(h0[k]) == 4
        assert abs(h1["train_loss"][-1] - h0["train_loss"][-1]) <= 0.25 * abs(h0["train_loss"][-1])
        assert h1["train_loss"][-1] < h1["train_loss"][0] and np.isfinite(s1).all()
    else:
        for k in h0:
            assert h1[k] == h0[k], k          # number for number
        np.testing.assert_array_equal(s1, s0)
