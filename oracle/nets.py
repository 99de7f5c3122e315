"""CPU oracle: a plain-torch restatement of the reference's FastSCNN / ContextNet path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``torch_semantic_segmentation_amd/`` may
import this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and only as the checker.

Parity pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference itself,
imported from /root/reference in the build container; see
``tests/golden/make_golden.py`` (generator, committed) and
``tests/test_oracle_golden.py`` (oracle == committed vectors, bit for bit where
the fixture holds full tensors).

What is restated (reference file:line, TSS = torch_semantic_segmentation/):

* conv -> batch-norm -> [relu] unit ......... TSS/models/fastscnn.py:164-185,
                                              TSS/models/contextnet.py:150-177
* depthwise -> bn -> pointwise -> bn -> [relu] TSS/models/fastscnn.py:188-199
* inverted-residual bottleneck ............... TSS/models/fastscnn.py:138-161,
                                              TSS/models/contextnet.py:129-147
* pyramid pooling ............................ TSS/models/fastscnn.py:101-123
* feature fusion (both flavours) ............. TSS/models/fastscnn.py:67-89,
                                              TSS/models/contextnet.py:104-126
* classifier heads ........................... TSS/models/fastscnn.py:92-98,
                                              TSS/models/contextnet.py:79-87
* whole nets ................................. TSS/models/fastscnn.py:15-64,
                                              TSS/models/contextnet.py:28-76

The module tree is generated from compact tables so that ``state_dict()`` keys,
shapes and dtypes equal the reference's (266 keys FastSCNN, 314 ContextNet) and
the ATen ops run in the reference's order, which makes the CPU results
bit-identical to an import of the reference in the same container.
"""
import torch
from torch import nn
from torch.nn import functional as F


def unit(cin, cout, k=1, stride=1, dilation=1, depthwise=False, act=True):
    """conv(bias=False) -> BatchNorm2d -> optional ReLU; padding = dilation for 3x3, 0 for 1x1."""
    pad = dilation if k == 3 else 0
    mods = [nn.Conv2d(cin, cout, k, stride=stride, padding=pad, dilation=dilation,
                      groups=cin if depthwise else 1, bias=False),
            nn.BatchNorm2d(cout)]
    if act:
        mods.append(nn.ReLU(inplace=True))
    return nn.Sequential(*mods)


def separable(cin, cout, stride=1):
    """FastSCNN flavour: depthwise and pointwise live in ONE Sequential, no ReLU between them."""
    dw = unit(cin, cin, 3, stride=stride, depthwise=True, act=False)
    pw = unit(cin, cout, 1)
    return nn.Sequential(*dw, *pw)


class InvertedResidual(nn.Module):
    """1x1 expand -> 3x3 depthwise -> 1x1 project; shape-equal skip; ReLU after the sum."""

    def __init__(self, cin, cout, stride=1, expansion=6):
        super().__init__()
        mid = cin * expansion
        self.conv1 = unit(cin, mid, 1)
        self.conv2 = unit(mid, mid, 3, stride=stride, depthwise=True)
        self.conv3 = unit(mid, cout, 1, act=False)

    def forward(self, x):
        y = self.conv3(self.conv2(self.conv1(x)))
        return y, x


class _FastResidual(InvertedResidual):
    def forward(self, x):                       # TSS/models/fastscnn.py:158-161 (x = x + input)
        y, x = super().forward(x)
        if y.shape == x.shape:
            y = y + x
        return F.relu(y)


class _CtxResidual(InvertedResidual):
    def forward(self, x):                       # TSS/models/contextnet.py:145-147 (x = input + x)
        y, x = super().forward(x)
        if y.shape == x.shape:
            y = x + y
        return F.relu(y)


def stack(block, cin, cout, repeats, stride, expansion=6):
    blocks = [block(cin, cout, stride=stride, expansion=expansion)]
    blocks += [block(cout, cout, expansion=expansion) for _ in range(repeats - 1)]
    return nn.Sequential(*blocks)


def up(x, size=None, scale=None):
    return F.interpolate(x, size=size, scale_factor=scale, mode='bilinear', align_corners=True)


class Pyramid(nn.Module):
    def __init__(self, cin, cout, bins=(1, 2, 3, 6)):
        super().__init__()
        self.pyramids = nn.ModuleList(
            [nn.Sequential(nn.AdaptiveAvgPool2d(b), unit(cin, cin // len(bins), 1)) for b in bins])
        self.conv = unit(cin * 2, cout, 1)

    def forward(self, x):
        branches = [up(p(x), size=x.shape[2:]) for p in self.pyramids.children()]
        return self.conv(torch.cat([x, *branches], dim=1))


class FastFusion(nn.Module):
    def __init__(self, low, high, cout, scale):
        super().__init__()
        self.lowres = nn.Sequential(
            nn.UpsamplingBilinear2d(scale_factor=scale),
            unit(low, low, 3, dilation=scale, depthwise=True),
            unit(low, cout, 1, act=False))
        self.highres = nn.Sequential(unit(high, cout, 1, act=False))

    def forward(self, low, high):
        low = self.lowres(low)
        high = self.highres(high)
        return F.relu(low + high)


class CtxFusion(nn.Module):
    def __init__(self, low, high, cout):
        super().__init__()
        self.lowres = nn.Sequential(unit(low, low, 3, dilation=4, depthwise=True),
                                    unit(low, cout, 1, act=False))
        self.highres = unit(high, cout, 1, act=False)

    def forward(self, low, high):
        low = self.lowres(up(low, size=high.shape[2:]))
        high = self.highres(high)
        return F.relu(low + high)


class FastSCNNOracle(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.downsample = nn.Sequential(unit(in_channels, 32, 3, stride=2),
                                        separable(32, 48, stride=2),
                                        separable(48, 64, stride=2))
        self.features = nn.Sequential(stack(_FastResidual, 64, 64, 3, 2),
                                      stack(_FastResidual, 64, 96, 3, 2),
                                      stack(_FastResidual, 96, 128, 3, 1),
                                      Pyramid(128, 128))
        self.fusion = FastFusion(128, 64, 128, 4)
        self.classifier = fast_head(128, out_channels)

    def forward(self, x):
        d = self.downsample(x)
        f = self.features(d)
        return up(self.classifier(self.fusion(f, d)), scale=8)


def fast_head(cin, cout):
    return nn.Sequential(separable(cin, cin), separable(cin, cin),
                         nn.Dropout(0.1), nn.Conv2d(cin, cout, kernel_size=1))


def ctx_head(cin, cout):
    return nn.Sequential(unit(cin, cin, 3, depthwise=True), unit(cin, cin, 1),
                         unit(cin, cin, 3, depthwise=True), unit(cin, cin, 1),
                         nn.Dropout(p=0.1), nn.Conv2d(cin, cout, 1))


class ContextNetOracle(nn.Module):
    def __init__(self, in_channels, out_channels, scale_factor=4):
        super().__init__()
        self.scale_factor = scale_factor
        self.spatial = nn.Sequential(
            unit(in_channels, 32, 3, stride=2),
            unit(32, 32, 3, stride=2, depthwise=True), unit(32, 64, 1),
            unit(64, 64, 3, stride=2, depthwise=True), unit(64, 128, 1),
            unit(128, 128, 3, depthwise=True), unit(128, 128, 1))
        self.context = nn.Sequential(
            unit(in_channels, 32, 3, stride=2),
            _CtxResidual(32, 32, expansion=1),
            _CtxResidual(32, 32, expansion=6),
            stack(_CtxResidual, 32, 48, 3, 2),
            stack(_CtxResidual, 48, 64, 3, 2),
            stack(_CtxResidual, 64, 96, 2, 1),
            stack(_CtxResidual, 96, 128, 2, 1),
            unit(128, 128, 3))
        self.feature_fusion = CtxFusion(128, 128, 128)
        self.classifier = ctx_head(128, out_channels)

    def forward(self, x):
        s = self.spatial(x)
        c = self.context(up(x, scale=1 / self.scale_factor))
        return up(self.classifier(self.feature_fusion(c, s)), scale=8)


def build(name, in_channels=3, out_channels=19):
    """name in {'fastscnn', 'contextnet12', 'contextnet14', 'contextnet18'}."""
    if name == 'fastscnn':
        return FastSCNNOracle(in_channels, out_channels)
    scale = {'contextnet12': 2, 'contextnet14': 4, 'contextnet18': 8}[name]
    return ContextNetOracle(in_channels, out_channels, scale_factor=scale)
