"""BiSeNet's attention blocks on the MI355X HIP path: `AttentionRefinementModule` and
`FeatureFusionModule` of TSS/models/bisenet.py:112-148 with the same constructor arguments, module tree and state_dict keys.
The global pool runs on the row-sliced pooling kernel, the 1x1 convolutions on the [B, C, 1, 1] pooled maps on the pointwise
kernels, and sigmoid + broadcast multiply (+ 1) in one pass (ops.gate, csrc/gate.hip).  BiSeNet's ResNet backbone is outside
the scope of this repository, so the whole model is not mirrored."""
from torch import nn

from .. import ops
from ._fused import FusedSequential, run

__all__ = ['AttentionRefinementModule', 'FeatureFusionModule', 'ConvBlock']


def ConvBlock(in_channels, out_channels, kernel_size, padding=0, stride=1, use_relu=True):
    """conv -> BN -> [ReLU]  (TSS/models/bisenet.py:151-160)"""
    layers = [nn.Conv2d(in_channels, out_channels, kernel_size, padding=padding, stride=stride, bias=False),
              nn.BatchNorm2d(out_channels)]
    if use_relu:
        layers += [nn.ReLU(inplace=True)]
    return FusedSequential(*layers)


class FeatureFusionModule(nn.Module):
    """(TSS/models/bisenet.py:112-131) x = ConvBlock3x3(input); x * (1 + sigmoid(conv(ConvBlock1x1(pool(x)))))"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = ConvBlock(in_channels, out_channels, 3, padding=1)
        self.attention = FusedSequential(
            nn.AdaptiveAvgPool2d(1),
            ConvBlock(out_channels, out_channels, 1),
            nn.Conv2d(out_channels, out_channels, 1),
            nn.Sigmoid())

    def forward(self, input):
        x = ops.materialize(run(self.conv, input))
        mods = list(self.attention)
        a = ops.adaptive_avg_pool(x, 1)
        a = run(mods[1], ops.Deferred(a))                       # 1x1 conv + BatchNorm + ReLU on the pooled [B, C, 1, 1] map
        a = ops.conv_unit(a, mods[2])                           # biased 1x1 conv; the sigmoid rides in the gate
        return ops.gate(x, ops.materialize(a), add_one=True)


class AttentionRefinementModule(nn.Module):
    """(TSS/models/bisenet.py:134-148) sigmoid(conv(pool(input))) * input"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        if in_channels != out_channels:
            raise ValueError("input and output channels must match")
        self.pool = nn.AdaptiveAvgPool2d(1)
        self.conv = nn.Conv2d(in_channels, out_channels, 1)
        self.activation = nn.Sigmoid()

    def forward(self, input):
        x = ops.to_nhwc(ops.materialize(input))
        a = ops.conv_unit(ops.adaptive_avg_pool(x, 1), self.conv)
        return ops.gate(x, ops.materialize(a), add_one=False)
