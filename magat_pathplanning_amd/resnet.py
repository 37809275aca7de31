"""Parameter containers for the per-agent CNN encoders with the reference's attribute names, so that
`state_dict()` keys/shapes equal the reference's (graphs/models/resnet_pytorch.py:40-73 BasicBlock,
334-425 ResNetSlim, 427-524 ResNet).  The eval-mode forward of these modules is NOT what inference
runs -- DecentralPlannerGATNet folds their parameters into the gfx950 encoder pack (encoder.py); the
torch forward below exists for training (autograd) only.
"""
import math

import torch.nn as nn


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False),
                                            nn.BatchNorm2d(planes))
        self.stride = stride

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        res = x if self.downsample is None else self.downsample(x)
        return self.relu(out + res)


class _ResNetBase(nn.Module):
    def __init__(self, channels, num_classes=128, pool_size=2):
        super().__init__()
        self.conv1 = nn.Conv2d(3, channels[0], 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(channels[0])
        self.relu = nn.ReLU(inplace=True)
        inplanes = channels[0]
        strides = [2] + [1] * (len(channels) - 1)
        for i, (c, s) in enumerate(zip(channels, strides)):
            setattr(self, "layer%d" % (i + 1), nn.Sequential(BasicBlock(inplanes, c, s)))
            inplanes = c
        self.n_layers = len(channels)
        self.avgpool = nn.AvgPool2d(pool_size)
        self.fc = nn.Conv2d(inplanes, num_classes, 1, 1, 0, bias=True)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def forward(self, x):
        x = self.relu(self.bn1(self.conv1(x)))
        for i in range(self.n_layers):
            x = getattr(self, "layer%d" % (i + 1))(x)
        return self.fc(self.avgpool(x))


class ResNet(_ResNetBase):
    """ResNet(BasicBlock,[1,1,1]), channels 32/64/128 ('ResNetLarge*', resnet_pytorch.py:427-524)."""

    def __init__(self):
        super().__init__((32, 64, 128))


class ResNetSlim(_ResNetBase):
    """ResNetSlim(BasicBlock,[1,1]), channels 32/64 ('ResNetSlim*', resnet_pytorch.py:334-425)."""

    def __init__(self):
        super().__init__((32, 64))
