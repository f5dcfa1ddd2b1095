"""The oracle's ResNet-18 trunk against an INDEPENDENT implementation of the same published architecture (CPU; round 6).

The reference builds EmbeddingNet's trunk with `torchvision.models.resnet18(pretrained=True)` (/root/reference/models/triplet.py:16).
torchvision is a third-party module that is neither under /root/reference nor installed in this image, so the reference's trunk cannot be
run here and `oracle/torch_twin.py::resnet18_features` restates torchvision's published definition — until round 6 with nothing outside
this repository to hold it to.  Hugging Face `transformers` IS installed and carries its own ResNet (`transformers.models.resnet`, the
implementation behind the `microsoft/resnet-18` checkpoint, converted from the torchvision / timm weights): a different code base for the
same network (ResNet(BasicBlock, [2, 2, 2, 2]): 7x7/2 stem - BatchNorm - ReLU - 3x3/2 max-pool, four stages of two basic blocks with the
stride on the block's first 3x3 convolution and a 1x1/stride + BatchNorm shortcut where the shape changes, global average pool).

Here the product's `ResNet18Trunk` module tree (torchvision's state_dict keys, what a torchvision checkpoint loads into) is given random
weights AND random BatchNorm parameters / running statistics, the same tensors are loaded into `transformers.ResNetModel` through the
key map below, and the oracle must reproduce that model's pooled features and — after a train-mode pass — every BatchNorm running
statistic, in eval and in train mode.  The parameter count is torchvision's resnet18 without its classifier (11 689 512 - 513 000).

What this pins: the oracle's restatement (and, through tests/test_triplet_gpu.py, the HIP trunk) to an implementation the builder did not
write.  What it does not: torchvision 0.2.1 itself, and the pre-trained weights (no network).
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(os.path.dirname(HERE), "srl-zoo_amd"), os.path.dirname(HERE), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

transformers = pytest.importorskip("transformers")


def _hf_key(k):
    """torchvision resnet18 state_dict key -> transformers.ResNetModel state_dict key (None: no counterpart, the classifier)."""
    parts = k.split(".")
    norm = {"weight": "weight", "bias": "bias", "running_mean": "running_mean", "running_var": "running_var",
            "num_batches_tracked": "num_batches_tracked"}
    if parts[0] == "conv1":
        return "embedder.embedder.convolution.weight"
    if parts[0] == "bn1":
        return "embedder.embedder.normalization." + norm[parts[1]]
    if parts[0].startswith("layer"):
        stage, block = int(parts[0][5:]) - 1, int(parts[1])
        base = "encoder.stages.%d.layers.%d." % (stage, block)
        what = parts[2]
        if what in ("conv1", "conv2"):
            return base + "layer.%d.convolution.weight" % (0 if what == "conv1" else 1)
        if what in ("bn1", "bn2"):
            return base + "layer.%d.normalization.%s" % (0 if what == "bn1" else 1, norm[parts[3]])
        if what == "downsample":
            return base + ("shortcut.convolution.weight" if parts[3] == "0" else "shortcut.normalization." + norm[parts[4]])
    return None


def _models(seed):
    from transformers import ResNetConfig, ResNetModel
    from models.triplet import ResNet18Trunk
    torch.manual_seed(seed)
    trunk = ResNet18Trunk()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():  # BatchNorm parameters and running statistics away from their (1, 0, 0, 1) initialisation
        for m in trunk.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.2)
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
    cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[64, 128, 256, 512], depths=[2, 2, 2, 2], layer_type="basic",
                       hidden_act="relu", downsample_in_first_stage=False)
    hf = ResNetModel(cfg)
    sd = trunk.state_dict()
    mapped = OrderedDict()
    for k, v in sd.items():
        hk = _hf_key(k)
        if hk is not None:
            mapped[hk] = v.clone()
    assert sorted(mapped) == sorted(hf.state_dict()), "the key map covers the whole independent model"
    hf.load_state_dict(mapped)
    n_params = sum(v.numel() for k, v in sd.items() if not k.startswith("fc.") and "running_" not in k and "num_batches" not in k)
    assert n_params == sum(p.numel() for p in hf.parameters()) == 11689512 - 513000
    return trunk, hf


@pytest.mark.parametrize("training", [False, True], ids=["eval_bn", "train_bn"])
def test_oracle_trunk_is_the_independent_resnet18(training):
    from oracle import torch_twin as T
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    trunk, hf = _models(7)
    x = torch.from_numpy(np.random.RandomState(3).randn(3, 3, 224, 224).astype(np.float32))
    state = OrderedDict(("model.conv_layers." + k, v.detach().clone()) for k, v in trunk.state_dict().items())
    hf.train(training)
    with torch.no_grad():
        want = hf(pixel_values=x).pooler_output.reshape(3, -1)
    got = T.resnet18_features(state, x, training)
    assert got.shape == want.shape == (3, 512)
    err = float((got - want).abs().max() / want.abs().max())
    assert err < 1e-5, err
    # the BatchNorm buffers after the pass: untouched in eval mode, one momentum update of the batch statistics in train mode
    after = hf.state_dict()
    moved = 0
    for k, v in trunk.state_dict().items():
        hk = _hf_key(k)
        if hk is None or ("running_" not in k and "num_batches" not in k):
            continue
        mine = state["model.conv_layers." + k]
        if "num_batches" in k:
            assert int(mine) == int(after[hk]) == (1 if training else 0), k
            continue
        e = float((mine.double() - after[hk].double()).abs().max() / max(float(after[hk].abs().max()), 1e-30))
        assert e < 1e-5, (k, e)
        moved += int(not torch.equal(after[hk], v))
    assert moved == (40 if training else 0)  # running_mean and running_var of the 20 BatchNorm layers
