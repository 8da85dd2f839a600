"""ResNet-18, CIFAR variant, state_dict-compatible with the reference.

The reference's classifier (Classification/models/ResNet.py:180-344) is torchvision's
ResNet with a 3x3 stride-1 stem, no max-pool and an input-normalisation layer inside the
network.  Only resnet18 is on the benchmarked path; it is rebuilt here
from a stage table so that `named_parameters()` yields the same 62 names / shapes / order
 — that order *is* the flat index the saliency ranking runs over —
and `state_dict()` the same 124 keys, so reference checkpoints and masks load unchanged.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

CIFAR10_MEAN = (0.4914, 0.4822, 0.4465)
CIFAR10_STD = (0.2470, 0.2435, 0.2616)


class NormalizeByChannelMeanStd(nn.Module):
    """(x - mean) / std per colour channel, as buffers `mean` / `std`
    (Classification/utils.py:297-318; installed on the model by setup_model_dataset)."""

    def __init__(self, mean, std):
        super().__init__()
        self.register_buffer("mean", torch.as_tensor(mean, dtype=torch.float32))
        self.register_buffer("std", torch.as_tensor(std, dtype=torch.float32))

    def forward(self, x):
        return x.sub(self.mean[None, :, None, None]).div(self.std[None, :, None, None])

    def extra_repr(self):
        return f"mean={self.mean}, std={self.std}"


class BasicBlock(nn.Module):
    """conv3x3-BN-ReLU-conv3x3-BN (+ 1x1 projection when the shape changes) + ReLU."""

    def __init__(self, cin: int, cout: int, stride: int):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        # registered after bn2 so the projection's parameters follow bn2 in named_parameters()
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    fused_bn = False  # set by unlearn_saliency_amd.norm.use_fused_bn
    fused_block = False  # set by use_fused_bn(model, blocks=True): the whole block as one autograd node

    def forward(self, x):
        if self.fused_block:
            from ...resblock import fused_basic_block
            out = fused_basic_block(self, x)
            if out is not None:
                return out
        if self.fused_bn:
            from ...norm import fused_bn_act
            y = fused_bn_act(self.conv1(x), self.bn1, relu=True)
            skip = x if self.downsample is None else fused_bn_act(self.downsample[0](x), self.downsample[1], relu=False)
            return fused_bn_act(self.conv2(y), self.bn2, residual=skip, relu=True)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        skip = x if self.downsample is None else self.downsample(x)
        return self.relu(y + skip)


class ResNetCifar(nn.Module):
    # (width, first-block stride) per stage; two BasicBlocks each for ResNet-18
    STAGES = ((64, 1), (128, 2), (256, 2), (512, 2))

    def __init__(self, num_classes: int = 10, blocks_per_stage=(2, 2, 2, 2), imagenet: bool = False):
        super().__init__()
        self.normalize = NormalizeByChannelMeanStd(CIFAR10_MEAN, CIFAR10_STD)
        if imagenet:
            self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
            self.maxpool = nn.MaxPool2d(3, 2, 1)
        else:
            self.conv1 = nn.Conv2d(3, 64, 3, 1, 1, bias=False)
            self.maxpool = nn.Identity()
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        cin = 64
        for i, ((width, stride), nblk) in enumerate(zip(self.STAGES, blocks_per_stage), start=1):
            stage = [BasicBlock(cin if b == 0 else width, width, stride if b == 0 else 1) for b in range(nblk)]
            setattr(self, f"layer{i}", nn.Sequential(*stage))
            cin = width
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(cin, num_classes)
        # same initialisation family as the reference (ResNet.py:245-250): Kaiming-normal
        # fan_out for convolutions, unit/zero affine for the norms; nn.Linear keeps its default.
        for mod in self.modules():
            if isinstance(mod, nn.Conv2d):
                nn.init.kaiming_normal_(mod.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(mod, nn.BatchNorm2d):
                nn.init.ones_(mod.weight)
                nn.init.zeros_(mod.bias)

    fused_bn = False

    def forward(self, x):
        x = self.normalize(x)
        if self.fused_bn:
            from ...norm import fused_bn_act
            x = self.maxpool(fused_bn_act(self.conv1(x), self.bn1, relu=True))
        else:
            x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def resnet18(num_classes: int = 10, imagenet: bool = False, **_unused) -> ResNetCifar:
    return ResNetCifar(num_classes=num_classes, blocks_per_stage=(2, 2, 2, 2), imagenet=imagenet)
