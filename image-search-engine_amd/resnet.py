"""ResNet-50 feature extractor in plain torch.nn (torchvision is not in this image).

The reference truncates torchvision's resnet50 at node ``flatten`` (2048-d,
backend/descriptors.py:161-168).  The layer shapes below follow the published
ResNet-50 v1.5 architecture that torchvision implements (stride on the 3x3
convolution of each bottleneck); parameter names match torchvision's state_dict
keys so IMAGENET1K_V2 weights load where a checkpoint is available offline
(``load_resnet50_weights``, ``CNNDescriptor(weights_path=...)``).  BASELINE config 2 uses seeded random-init weights (the
reference fetches its weights from the network, which is impossible here).
"""
from __future__ import annotations

import torch
import torch.nn as nn


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample: nn.Module | None = None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class ResNet50Features(nn.Module):
    """forward(x: (B,3,H,W) float) -> (B, 2048): the ``flatten`` node of resnet50."""

    def __init__(self):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, 3, 1)
        self.layer2 = self._make_layer(128, 4, 2)
        self.layer3 = self._make_layer(256, 6, 2)
        self.layer4 = self._make_layer(512, 3, 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        for m in self.modules():  # torchvision's default init
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes: int, blocks: int, stride: int) -> nn.Sequential:
        downsample = None
        if stride != 1 or self.inplanes != planes * Bottleneck.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * Bottleneck.expansion, 1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * Bottleneck.expansion),
            )
        layers = [Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * Bottleneck.expansion
        layers += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return torch.flatten(self.avgpool(x), 1)


def resnet50_features(seed: int | None = 0) -> ResNet50Features:
    """Seeded random-init extractor (BASELINE config 2)."""
    if seed is None:
        return ResNet50Features()
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(seed)
        return ResNet50Features()


def load_resnet50_weights(net: ResNet50Features, path) -> ResNet50Features:
    """Load a torchvision ResNet-50 ``state_dict`` (the file behind
    ``resnet50(weights=IMAGENET1K_V2)``, backend/descriptors.py:161-163) into the extractor, BEFORE
    BatchNorm is folded.  The file is read with ``weights_only=True`` (nothing in it is executed);
    the classifier head ``fc.*`` is not part of the ``flatten`` feature node and is dropped; any
    other missing or unexpected entry is an error."""
    state = torch.load(str(path), map_location="cpu", weights_only=True)
    if isinstance(state, dict) and "state_dict" in state and isinstance(state["state_dict"], dict):
        state = state["state_dict"]
    state = {k: v for k, v in state.items() if not k.startswith("fc.")}
    missing, unexpected = net.load_state_dict(state, strict=False)
    missing = [k for k in missing if not k.endswith("num_batches_tracked")]
    if missing or unexpected:
        raise RuntimeError(f"not a ResNet-50 state_dict: missing {missing[:5]}, unexpected {list(unexpected)[:5]}")
    return net


def fold_batchnorm_(net: ResNet50Features) -> ResNet50Features:
    """Inference-only: fold every BatchNorm into the convolution in front of it (eval-mode
    statistics), in place.  Same function up to fp32 rounding, one elementwise pass less per layer."""
    from torch.nn.utils.fusion import fuse_conv_bn_eval

    net.eval()
    net.conv1 = fuse_conv_bn_eval(net.conv1, net.bn1)
    net.bn1 = nn.Identity()
    for layer in (net.layer1, net.layer2, net.layer3, net.layer4):
        for blk in layer:
            blk.conv1, blk.bn1 = fuse_conv_bn_eval(blk.conv1, blk.bn1), nn.Identity()
            blk.conv2, blk.bn2 = fuse_conv_bn_eval(blk.conv2, blk.bn2), nn.Identity()
            blk.conv3, blk.bn3 = fuse_conv_bn_eval(blk.conv3, blk.bn3), nn.Identity()
            if blk.downsample is not None:
                blk.downsample = nn.Sequential(fuse_conv_bn_eval(blk.downsample[0], blk.downsample[1]))
    return net
