"""ASPP (atrous spatial pyramid pooling) decoder head on the MI355X HIP path -- BASELINE config 5 names a "DeepLab-style
ASPP/dilated-decoder head"; the reference has no such module (its dilated decoder is the pyramid
pooling + dilation-4 fusion of FastSCNN), so this one follows the published DeepLabV3 head (Chen et al. 2017; the layout
of torchvision.models.segmentation.deeplabv3.ASPP / DeepLabHead, whose state_dict keys it keeps) and its parity is
pinned to plain torch (oracle/aspp.py), NOT to the reference.

    ASPP(in, out, rates):  [1x1 conv-BN-ReLU] + [3x3 dilated conv-BN-ReLU per rate] + [global pool -> 1x1 conv-BN-ReLU ->
                           upsample] -> concat -> 1x1 conv-BN-ReLU -> Dropout(0.5)
    ASPPHead(in, classes): ASPP -> 3x3 conv-BN-ReLU -> 1x1 conv (bias)

Every branch is a deferred conv unit (ops.conv_unit: dense dilated 3x3 through tss_conv3x3_*, 1x1 through the pointwise
kernels); the branches' BatchNorm + ReLU are applied while they are written into their channel slice of the concat buffer
(ops.concat_joined), so no normalised branch tensor and no concat copy exist.
"""
import torch
from torch import nn

from .. import ops
from ._fused import Deferred, FusedSequential, HipModel, has_hooks, run
from .fastscnn import FastSCNN

__all__ = ['ASPP', 'ASPPHead', 'FastSCNNASPP', 'fastscnn_aspp']


def _conv_bn_relu(cin, cout, k, dilation=1):
    return FusedSequential(nn.Conv2d(cin, cout, k, padding=dilation if k == 3 else 0, dilation=dilation, bias=False),
                           nn.BatchNorm2d(cout), nn.ReLU())


class ASPP(nn.Module):
    def __init__(self, in_channels, out_channels=256, atrous_rates=(12, 24, 36), dropout=0.5):
        super().__init__()
        mods = [_conv_bn_relu(in_channels, out_channels, 1)]
        mods += [_conv_bn_relu(in_channels, out_channels, 3, dilation=r) for r in atrous_rates]
        mods.append(FusedSequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(in_channels, out_channels, 1, bias=False),
                                    nn.BatchNorm2d(out_channels), nn.ReLU()))
        self.convs = nn.ModuleList(mods)
        self.project = FusedSequential(nn.Conv2d(len(mods) * out_channels, out_channels, 1, bias=False),
                                       nn.BatchNorm2d(out_channels), nn.ReLU(), nn.Dropout(dropout))

    def forward(self, input):
        x = ops.to_nhwc(ops.materialize(input))
        branches = [run(m, x) for m in list(self.convs)[:-1]]
        pooled = run(self.convs[-1], x)                                        # (B, out, 1, 1), BatchNorm + ReLU pending
        if not self.training and not torch.is_grad_enabled() and not has_hooks(self.project) and not has_hooks(self.convs):
            # eval forward: the project layer reads the five branches in place, their BatchNorm + ReLU applied on load, the pooled
            # row broadcast (one image) -- no concat buffer, no upsampled copy of a 1 x 1 map (ops.conv_unit_multi)
            d = ops.conv_unit_multi(branches + [pooled], self.project[0], self.project[1], relu=True)
            if d is not None:
                return ops.materialize(d)              # (eval mode: the project's nn.Dropout is the identity)
        pooled = ops.materialize(pooled)
        branches.append(Deferred(ops.bilinear(pooled, size=tuple(x.shape[2:]))))   # align_corners is immaterial for a 1x1 source
        return self.project(ops.concat_joined(branches, relu=False))


class ASPPHead(FusedSequential):
    def __init__(self, in_channels, num_classes, atrous_rates=(12, 24, 36), mid_channels=256):
        super().__init__(ASPP(in_channels, mid_channels, atrous_rates),
                         nn.Conv2d(mid_channels, mid_channels, 3, padding=1, bias=False),
                         nn.BatchNorm2d(mid_channels), nn.ReLU(),
                         nn.Conv2d(mid_channels, num_classes, 1))


class FastSCNNASPP(HipModel):
    """FastSCNN's learning-to-downsample + global feature extractor + fusion (1/8 resolution, 128 channels) under a
    DeepLab-style ASPP head and the x8 bilinear upsample: the config-5 stress model (dilated dense 3x3 + upsample)."""

    logit_scale = 8

    def __init__(self, in_channels, out_channels, atrous_rates=(6, 12, 18), mid_channels=128):
        super().__init__()
        base = FastSCNN(in_channels, out_channels)
        self.downsample, self.features, self.fusion = base.downsample, base.features, base.fusion
        self.classifier = ASPPHead(128, out_channels, atrous_rates, mid_channels)

    def forward_lowres(self, input):
        downsample = self.downsample(self.image_in(input))
        features = self.features(downsample)
        return self.classifier(self.fusion(features, downsample))

    def forward(self, input):
        return self.logits_out(ops.upsample_logits(self.forward_lowres(input), scale_factor=self.logit_scale), input)


def fastscnn_aspp(in_channels, out_channels):
    return FastSCNNASPP(in_channels, out_channels)
