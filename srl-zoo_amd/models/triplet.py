"""EmbeddingNet — frozen ResNet-18 trunk + Linear(512,128) + PReLU + Linear(128,S) for the time-contrastive triplet loss
(reference models/triplet.py:6-39), MI355X-native.

The reference builds the trunk with `torchvision.models.resnet18(pretrained=True)` (torchvision 0.2.1, environment.yml:84),
freezes it (triplet.py:17-19) and replaces its `fc`.  torchvision is a third-party module that is NOT under /root/reference and
is not installed here, and the pre-trained weights cannot be downloaded: `ResNet18Trunk` below restates the canonical
torchvision ResNet-18 module tree (same sub-module names -> same state_dict keys, same construction order and
initialisers: kaiming_normal_(fan_out, relu) convolutions, BatchNorm weight 1 / bias 0), so a torchvision `resnet18` state_dict
loads as is (`SRLZ_RESNET18_WEIGHTS=/path/to/resnet18.pth`, else the random initialisation stays).  PARITY IS UNPINNED for
the trunk against the reference: there is no reference run to compare with, only the CPU oracle's restatement (oracle/torch_twin.py),
which tests/test_oracle_resnet_independent.py holds to Hugging Face transformers' independent ResNet-18 on the same weights.

The containers hold parameters only; the forward runs the HIP kernels (srlz/hotpath.py::resnet18_forward): forward-only —
nothing is back-propagated through a frozen trunk — with BatchNorm in whatever mode the module is in (the reference leaves
the frozen trunk in train() mode during training minibatches, models/learner.py:365, so its batch statistics are used and
its running statistics move).
"""
from __future__ import print_function, division, absolute_import

import os

import torch as th
import torch.nn as nn

from .models import BaseModelSRL
from srlz import hotpath, ops


def _conv3x3(inp, out, stride=1):
    return nn.Conv2d(inp, out, kernel_size=3, stride=stride, padding=1, bias=False)


class BasicBlock(nn.Module):
    """torchvision.models.resnet.BasicBlock (parameters only)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(BasicBlock, self).__init__()
        self.conv1 = _conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


class ResNet18Trunk(nn.Module):
    """torchvision.models.resnet18() module tree: conv1, bn1, layer1..4 ([2, 2, 2, 2] BasicBlocks), fc."""

    def __init__(self, num_classes=1000):
        super(ResNet18Trunk, self).__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, 2)
        self.layer2 = self._make_layer(128, 2, stride=2)
        self.layer3 = self._make_layer(256, 2, stride=2)
        self.layer4 = self._make_layer(512, 2, stride=2)
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(BasicBlock(self.inplanes, planes))
        return nn.Sequential(*layers)


class EmbeddingNet(BaseModelSRL):
    """ResNet-18 (frozen) + FC layers learning a metric embedding.
    input: 3-channel RGB observations [B, 3, 224, 224] (the anchor / positive / negative views one at a time)
    :param state_dim: (int)
    :param embedding_size: (int) size of the TCN embedding
    """

    def __init__(self, state_dim=2, embedding_size=128):
        super(EmbeddingNet, self).__init__()
        self.conv_layers = ResNet18Trunk()
        weights = os.environ.get("SRLZ_RESNET18_WEIGHTS", "")
        if weights:  # a torchvision resnet18 state_dict (what `pretrained=True` would have downloaded)
            self.conv_layers.load_state_dict(th.load(weights, map_location="cpu"))
        for param in self.conv_layers.parameters():
            param.requires_grad = False
        n_units = self.conv_layers.fc.in_features
        print("{} units in the last layer".format(n_units))
        self.conv_layers.fc = nn.Linear(n_units, embedding_size)
        self.fc = nn.Sequential(nn.PReLU(), nn.Linear(embedding_size, state_dim))

    def _head(self, feat):
        x = hotpath.linear(self.conv_layers.fc, feat)
        x = x.view(x.size(0), -1)
        x = ops.PReLUFn.apply(x, self.fc[0].weight)
        return hotpath.linear(self.fc[1], x)

    def forward(self, x):
        hotpath.require_gpu(x, "EmbeddingNet")
        feat = hotpath.resnet18_forward(self.conv_layers, x, self.conv_layers.training)  # [B, 512], no gradient
        return self._head(feat)

    def forwardViews(self, views, want=None):
        """[self(v) for v in views] with ONE pass of the frozen trunk over all the views batched along n, one BatchNorm group per
        view (hotpath.resnet18_forward(groups=len(views))): features, running statistics and num_batches_tracked are those of the
        separate calls in list order (the reference makes them one after the other, models/learner.py:383-391 via
        modules.py:92-100).  want[i] = False: the trunk still sees view i (its BatchNorm statistics move, as in the reference) but
        the head is not run for it and None is returned in its place."""
        hotpath.require_gpu(views[0], "EmbeddingNet")
        b = views[0].shape[0]
        assert all(v.shape == views[0].shape for v in views), "the views of one batched trunk pass have one shape"
        x = th.cat([ops.frames_as_float(v) for v in views], 0)
        feat = hotpath.resnet18_forward(self.conv_layers, x, self.conv_layers.training, groups=len(views))
        return [self._head(feat[i * b:(i + 1) * b]) if (want is None or want[i]) else None for i in range(len(views))]

    def getStates(self, observations):
        """For inference the forward pass is done on the positive observation (first view)."""
        return self.forward(observations[:, :3:, :, :])
