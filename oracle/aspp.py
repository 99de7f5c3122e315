"""CPU oracle, part 4 (TEST INFRASTRUCTURE ONLY): plain-torch ASPP head and PSPNet restatement.

* ASPP / ASPPHead: the reference has NO ASPP module; BASELINE config 5 asks for a DeepLab-style
  one, so this is the published DeepLabV3 head (Chen et al. 2017, layout and state_dict keys of torchvision's
  deeplabv3.ASPP / DeepLabHead) written with torch.nn leaves.  Parity of the product's ASPP is therefore pinned to torch,
  not to the reference -- the tests say so.
* pyramid_pools / PSPHead: restates TSS/models/pspnet.py:6-58 (pinned to the imported reference by
  tests/golden/pspnet.npz, tests/test_oracle_golden.py).
"""
import torch
from torch import nn
from torch.nn import functional as F

from . import nets


def _cbr(cin, cout, k, dilation=1):
    return nn.Sequential(nn.Conv2d(cin, cout, k, padding=dilation if k == 3 else 0, dilation=dilation, bias=False),
                         nn.BatchNorm2d(cout), nn.ReLU())


class ASPP(nn.Module):
    def __init__(self, in_channels, out_channels=256, atrous_rates=(12, 24, 36), dropout=0.5):
        super().__init__()
        mods = [_cbr(in_channels, out_channels, 1)] + [_cbr(in_channels, out_channels, 3, r) for r in atrous_rates]
        mods.append(nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(in_channels, out_channels, 1, bias=False),
                                  nn.BatchNorm2d(out_channels), nn.ReLU()))
        self.convs = nn.ModuleList(mods)
        self.project = nn.Sequential(nn.Conv2d(len(mods) * out_channels, out_channels, 1, bias=False),
                                     nn.BatchNorm2d(out_channels), nn.ReLU(), nn.Dropout(dropout))

    def forward(self, x):
        outs = [m(x) for m in list(self.convs)[:-1]]
        outs.append(F.interpolate(self.convs[-1](x), size=x.shape[-2:], mode='bilinear', align_corners=False))
        return self.project(torch.cat(outs, dim=1))


class ASPPHead(nn.Sequential):
    def __init__(self, in_channels, num_classes, atrous_rates=(12, 24, 36), mid_channels=256):
        super().__init__(ASPP(in_channels, mid_channels, atrous_rates),
                         nn.Conv2d(mid_channels, mid_channels, 3, padding=1, bias=False),
                         nn.BatchNorm2d(mid_channels), nn.ReLU(),
                         nn.Conv2d(mid_channels, num_classes, 1))


class FastSCNNASPP(nn.Module):
    def __init__(self, in_channels, out_channels, atrous_rates=(6, 12, 18), mid_channels=128):
        super().__init__()
        base = nets.FastSCNNOracle(in_channels, out_channels)
        self.downsample, self.features, self.fusion = base.downsample, base.features, base.fusion
        self.classifier = ASPPHead(128, out_channels, atrous_rates, mid_channels)

    def forward(self, x):
        d = self.downsample(x)
        f = self.features(d)
        return nets.up(self.classifier(self.fusion(f, d)), scale=8)


class PyramidPools(nn.ModuleList):
    """TSS/models/pspnet.py:26-58."""

    def __init__(self, in_channels, out_channels, pools=(1, 2, 3, 6)):
        if out_channels % len(pools) != 0:
            raise ValueError("output channels must be divisible by the number of pools")
        pc = out_channels // len(pools)
        super().__init__([nn.Sequential(nn.AdaptiveAvgPool2d(b), nn.Sequential(nn.Conv2d(in_channels, pc, 1, bias=False),
                                                                               nn.BatchNorm2d(pc), nn.ReLU(inplace=True)))
                          for b in pools])

    def forward(self, x):
        return torch.cat([nets.up(p(x), size=x.shape[2:]) for p in self.children()], dim=1)


class PSPNetOracle(nn.Module):
    """TSS/models/pspnet.py:6-23."""

    def __init__(self, backbone, out_channels, feature_channels):
        super().__init__()
        self.backbone = backbone
        self.ppm = PyramidPools(feature_channels, feature_channels)
        self.classifier = nn.Conv2d(feature_channels * 2, out_channels, 1)

    def forward(self, x):
        feat = self.backbone(x)
        return self.classifier(torch.cat([feat, self.ppm(feat)], dim=1))


# ----------------------------------------------------------------------------- LEDNet split-shuffle unit (TSS/models/lednet.py:95-124,157-188)

def factorized(channels, dilation=1, act=True):
    mods = [nn.Conv2d(channels, channels, (1, 3), padding=(0, dilation), dilation=(1, dilation), bias=False), nn.ReLU(inplace=True),
            nn.Conv2d(channels, channels, (3, 1), padding=(dilation, 0), dilation=(dilation, 1), bias=False), nn.BatchNorm2d(channels)]
    if act:
        mods.append(nn.ReLU(inplace=True))
    return nn.Sequential(*mods)


def shuffle(x, groups):
    b, c, h, w = x.shape
    return x.reshape(b, groups, c // groups, h, w).transpose(1, 2).reshape(b, c, h, w)


class SSnbt(nn.Module):
    def __init__(self, channels, dilation=1, dropout_p=0.0):
        super().__init__()
        half = channels // 2
        self.left = nn.Sequential(factorized(half), factorized(half, dilation, act=False))
        self.right = nn.Sequential(factorized(half), factorized(half, dilation, act=False))
        self.activation = nn.ReLU(inplace=True)
        self.dropout = nn.Dropout2d(p=dropout_p)

    def forward(self, x):
        left, right = torch.chunk(x, 2, 1)
        y = torch.cat([self.left(left), self.right(right)], dim=1)
        y = self.dropout(y)
        return shuffle(self.activation(x + y), 2)


# ---- BiSeNet's attention blocks (TSS/models/bisenet.py:112-148), pinned to the imported reference by tests/golden/pspnet.npz
class ChannelGateFusion(nn.Module):
    """FeatureFusionModule, TSS/models/bisenet.py:112-131: a 3x3 conv block, then the map is scaled by 1 + a sigmoid gate computed
    from its own global average (pool -> 1x1 conv block -> biased 1x1 conv -> sigmoid).  Same attribute names / keys."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_channels, out_channels, 3, padding=1, bias=False), nn.BatchNorm2d(out_channels),
                                  nn.ReLU(inplace=True))
        self.attention = nn.Sequential(
            nn.AdaptiveAvgPool2d(1),
            nn.Sequential(nn.Conv2d(out_channels, out_channels, 1, bias=False), nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True)),
            nn.Conv2d(out_channels, out_channels, 1),
            nn.Sigmoid())

    def forward(self, x):
        y = self.conv(x)
        return y * (1. + self.attention(y))


class ChannelGateRefine(nn.Module):
    """AttentionRefinementModule, TSS/models/bisenet.py:134-148: the input scaled by a sigmoid gate from its global average."""

    def __init__(self, channels):
        super().__init__()
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.conv = nn.Conv2d(channels, channels, 1)
        self.activation = nn.Sigmoid()

    def forward(self, x):
        return self.activation(self.conv(self.pool(x))) * x
