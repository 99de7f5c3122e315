"""PSPNet head on the MI355X HIP path: the same public surface, module tree and state_dict keys as
TSS/models/pspnet.py (`PSPNet`, `PyramidPoolingModule`); the arithmetic runs in the pyramid kernels of the FastSCNN path
(tss_ppm_pool_*, tss_ppm_concat_*: every arm per launch, BatchNorm + ReLU applied per bilinear tap, in-place concat) and
the 1x1 convolution kernels behind include/tss_hip.h -- never in ATen.  The backbone is whatever module the caller passes
(TSS/models/pspnet.py:8-12); it must return a CUDA feature map with a multiple of 8 channels.
"""
from torch import nn

from .. import ops
from ._fused import Deferred, FusedSequential, HipModel, has_hooks, run

__all__ = ['PSPNet', 'PyramidPoolingModule']


class PyramidPoolingModule(nn.ModuleList):
    """(TSS/models/pspnet.py:26-58) children: Sequential(AdaptiveAvgPool2d(bin), Sequential(Conv2d 1x1, BatchNorm2d, ReLU));
    forward returns cat([upsample(arm(x)) for arm], 1) -- WITHOUT x, as the reference does."""

    def __init__(self, in_channels, out_channels, pools=[1, 2, 3, 6]):
        if out_channels % len(pools) != 0:
            raise ValueError("output channels must be divisible by the number of pools")
        pool_channels = out_channels // len(pools)
        arms = [FusedSequential(nn.AdaptiveAvgPool2d(pool_size),
                                FusedSequential(nn.Conv2d(in_channels, pool_channels, 1, bias=False),
                                                nn.BatchNorm2d(pool_channels), nn.ReLU(inplace=True)))
                for pool_size in pools]
        super().__init__(arms)

    def concat_with(self, x):
        """cat([x, *upsampled arms], 1): what PSPNet.forward needs (TSS/models/pspnet.py:21-23), in one buffer."""
        x = ops.to_nhwc(ops.materialize(x))
        arms = list(self.children())
        plain = ops.ppm_fused and all(
            isinstance(a, FusedSequential) and len(a) == 2 and isinstance(a[0], nn.AdaptiveAvgPool2d)
            and isinstance(a[0].output_size, int) and not has_hooks(a) for a in arms)
        if plain and len(arms) <= 4:
            pooled = ops.adaptive_avg_pool_multi(x, [a[0].output_size for a in arms])
            ds = [run(a[1], Deferred(p)) for a, p in zip(arms, pooled)]
            if ops.ppm_arms_fusable(x, ds):
                return ops.concat_upsampled_arms(x, ds)
            return ops.concat_upsampled(x, ds)
        return ops.concat_upsampled(x, [arm(x) for arm in arms])

    def forward(self, input):
        c = input.shape[1]
        return self.concat_with(input)[:, c:]


class PSPNet(HipModel):
    """(TSS/models/pspnet.py:6-23)"""

    def __init__(self, backbone, out_channels, feature_channels):
        super().__init__()
        self.backbone = backbone
        self.ppm = PyramidPoolingModule(feature_channels, feature_channels, pools=[1, 2, 3, 6])
        self.classifier = FusedClassifier(feature_channels * 2, out_channels)

    def forward(self, input):
        feat = self.backbone(self.image_in(input))
        if has_hooks(self.ppm):      # a hook on the pyramid must see what the reference's module returns: the pools alone
            feat = ops.to_nhwc(ops.materialize(feat))
            x = ops.concat([feat, self.ppm(feat)])
        else:
            x = self.ppm.concat_with(feat)
        return self.logits_out(self.classifier(x), input)


class FusedClassifier(nn.Conv2d):
    """nn.Conv2d(feature_channels * 2, out_channels, 1) of TSS/models/pspnet.py:17 (same parameters and state_dict keys
    `classifier.weight` / `classifier.bias`), executed by the HIP 1x1 kernels; returns NCHW-contiguous logits."""

    def __init__(self, in_channels, out_channels):
        super().__init__(in_channels, out_channels, 1)

    def forward(self, input):
        d = ops.conv_unit(input, self, None, False)
        return ops.materialize(d).contiguous()
