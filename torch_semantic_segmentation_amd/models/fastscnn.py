"""FastSCNN on the MI355X HIP path: the same public surface as TSS/models/fastscnn.py
(`FastSCNN`, `fastscnn`, and the block builders the training script imports, scripts/train_fastscnn.py:28),
the same module tree / state_dict keys (266), every container hookable; the arithmetic runs in the HIP
kernels behind include/tss_hip.h (ops.py), never in ATen.
"""
import torch
from torch import nn

from .. import ops
from ._fused import Deferred, FusedSequential, HipModel, has_hooks, run

__all__ = ['FastSCNN', 'fastscnn']


def fastscnn(in_channels, out_channels):
    return FastSCNN(in_channels, out_channels)


def _conv_bn(cin, cout, kernel_size, stride, padding, dilation, groups, use_activation):
    mods = [nn.Conv2d(cin, cout, kernel_size, stride=stride, padding=padding, dilation=dilation,
                      groups=groups, bias=False),
            nn.BatchNorm2d(cout)]
    if use_activation:
        mods.append(nn.ReLU(inplace=True))
    return mods


def Conv2dBlock(in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, use_activation=True):
    """conv -> BN -> [ReLU]  (TSS/models/fastscnn.py:164-173)"""
    return FusedSequential(*_conv_bn(in_channels, out_channels, kernel_size, stride, padding, dilation, 1,
                                     use_activation))


def DWConv2dBlock(in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, use_activation=True):
    """depthwise conv -> BN -> [ReLU]  (TSS/models/fastscnn.py:176-185)"""
    return FusedSequential(*_conv_bn(in_channels, out_channels, kernel_size, stride, padding, dilation,
                                     in_channels, use_activation))


def DSConv2dBlock(in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, use_activation=True):
    """depthwise -> BN -> pointwise -> BN -> [ReLU], no non-linearity in between (TSS/models/fastscnn.py:188-199)"""
    return FusedSequential(
        *_conv_bn(in_channels, in_channels, kernel_size, stride, padding, dilation, in_channels, False),
        *_conv_bn(in_channels, out_channels, 1, 1, 0, 1, 1, use_activation))


class BottleneckBlock(nn.Module):
    """1x1 expand -> dw 3x3 -> 1x1 project, shape-equal skip, ReLU after the sum (TSS/models/fastscnn.py:138-161).
    The two expanded tensors stay raw in HBM; only the block output is normalised and written."""

    def __init__(self, in_channels, out_channels, stride=1, expansion=6):
        super().__init__()
        mid = expansion * in_channels
        self.conv1 = Conv2dBlock(in_channels, mid, kernel_size=1)
        self.conv2 = DWConv2dBlock(mid, mid, kernel_size=3, padding=1, stride=stride)
        self.conv3 = Conv2dBlock(mid, out_channels, kernel_size=1, use_activation=False)

    def forward(self, input):
        x = ops.to_nhwc(ops.materialize(input))
        # shape-equal skip: the block input has two consumers; their gradients meet in the epilogue of conv1's backward-data kernel
        # (ops.residual_fork) instead of in an elementwise launch of autograd's
        if not (has_hooks(self.conv1) or has_hooks(self.conv2) or has_hooks(self.conv3)):
            # model.eval(), no gradient, bf16: the whole block as one kernel, the 6x-expanded tensors in LDS only (csrc/bneck.hip)
            out = ops.bottleneck_eval(x, self.conv1, self.conv2, self.conv3)
            if out is not None:
                return out
        c1, c2, c3 = self.conv1[0], self.conv2[0], self.conv3[0]
        residual = tuple(c2.stride) == (1, 1) and c1.in_channels == c3.out_channels and not has_hooks(self.conv1)
        xa, xb, fork = ops.residual_fork(x) if residual else (x, x, None)
        try:
            d = run(self.conv2, run(self.conv1, xa))
        finally:                                   # (also when a layer raises: ids are recycled, a stale entry would be picked up later)
            if fork is not None:
                ops._pending_forks.pop(id(xa), None)
        if fork is None and not has_hooks(self.conv3):
            # model.eval(), no gradient: the frozen BatchNorm, the skip and the ReLU ride in conv3's epilogue -- no join pass
            same_shape = (d.shape[0], c3.out_channels) + tuple(d.shape[2:]) == tuple(x.shape)
            out = ops.conv_unit_joined(d, self.conv3, xb if same_shape else None, relu=True)
            if out is not None:
                return out
        d = run(self.conv3, d)
        same = tuple(d.shape) == tuple(x.shape)
        return ops.join(d, xb if same else None, relu=True, fork=fork if same else None)


def BottleneckModule(in_channels, out_channels, expansion, repeats=1, stride=1):
    blocks = [BottleneckBlock(in_channels, out_channels, expansion=expansion, stride=stride)]
    blocks += [BottleneckBlock(out_channels, out_channels, expansion=expansion) for _ in range(1, repeats)]
    return FusedSequential(*blocks)


class PyramidPoolingModule(nn.Module):
    """(TSS/models/fastscnn.py:101-123) pooled branches are written straight into the concat buffer."""

    def __init__(self, in_channels, out_channels, pyramids=(1, 2, 3, 6)):
        super().__init__()
        self.pyramids = nn.ModuleList([
            FusedSequential(nn.AdaptiveAvgPool2d(bins),
                            Conv2dBlock(in_channels, in_channels // len(pyramids), kernel_size=1))
            for bins in pyramids])
        self.conv = Conv2dBlock(in_channels * 2, out_channels, kernel_size=1)

    def forward(self, input):
        x = ops.to_nhwc(ops.materialize(input))
        arms = list(self.pyramids.children())
        # every arm is (AdaptiveAvgPool2d(bins), Conv2dBlock): pools of all arms from one launch, each arm's 1x1 unit on
        # its pooled map, then BN + ReLU + upsample + concat of all arms in one launch (53 -> 25 launches per train step)
        plain = ops.ppm_fused and all(
            isinstance(a, FusedSequential) and len(a) == 2 and isinstance(a[0], nn.AdaptiveAvgPool2d)
            and isinstance(a[0].output_size, int) and not has_hooks(a) for a in arms)
        if plain:
            # x feeds the pools and the concat: the concat's gradient of x is added inside the pools' backward kernel (ops.fork_two)
            xp, xc, fork = ops.fork_two(x)
            try:
                pooled = ops.adaptive_avg_pool_multi(xp, [a[0].output_size for a in arms])
                x = xc
                # the arms' 1x1 convolutions + BatchNorm statistics: one launch for all arms (csrc/ppm.hip), else unit by unit
                ds = ops.ppm_arms([a[1] for a in arms], pooled)
                if ds is None:
                    ds = [run(a[1], Deferred(p)) for a, p in zip(arms, pooled)]
                if ops.ppm_arms_fusable(x, ds):
                    cat = ops.concat_upsampled_arms(x, ds)
                else:
                    cat = ops.concat_upsampled(x, ds)
            finally:
                ops.drop_fork(xp, xc)
            return self.conv(cat)
        pools = [pool(x) for pool in arms]
        return self.conv(ops.concat_upsampled(x, pools))


class FeatureFusionModule(nn.Module):
    """(TSS/models/fastscnn.py:67-89) relu(lowres + highres) with both BatchNorms applied inside the join."""

    def __init__(self, in_channels, out_channels, scale_factor):
        super().__init__()
        lowres_channels, highres_channels = in_channels
        self.lowres = FusedSequential(
            nn.UpsamplingBilinear2d(scale_factor=scale_factor),
            DWConv2dBlock(lowres_channels, lowres_channels, kernel_size=3, padding=scale_factor,
                          dilation=scale_factor),
            Conv2dBlock(lowres_channels, out_channels, kernel_size=1, use_activation=False))
        self.highres = FusedSequential(
            Conv2dBlock(highres_channels, out_channels, kernel_size=1, use_activation=False))

    def forward(self, lowres, highres):
        if not torch.is_grad_enabled() and not (has_hooks(self.lowres) or has_hooks(self.highres)) and len(self.lowres) == 3 \
                and len(self.highres) == 1:
            # model.eval(), no gradient: no join pass -- the low-resolution branch's last layer writes BN(conv) itself, the high-resolution
            # layer adds it and applies the ReLU in its own epilogue (tss_pwconv_fwd_joined): two writes + one read instead of
            # two writes + two reads + one write
            d2 = self.lowres.unit(lowres, upto=2)
            low = ops.conv_unit_joined(d2, self.lowres[2], None, relu=False)
            if low is None:                    # outside the fused epilogue's envelope (f32, channel counts): the ordinary chain
                return ops.join(run(self.lowres[2], d2), run(self.highres, highres), relu=True)
            out = ops.conv_unit_joined(highres, self.highres[0], low, relu=True)
            return out if out is not None else ops.join(low, run(self.highres, highres), relu=True)
        return ops.join(run(self.lowres, lowres), run(self.highres, highres), relu=True)


def Classifier(in_channels, out_channels):
    """(TSS/models/fastscnn.py:92-98) also used for the deep-supervision heads of the training script."""
    return FusedSequential(
        DSConv2dBlock(in_channels, in_channels, kernel_size=3, padding=1),
        DSConv2dBlock(in_channels, in_channels, kernel_size=3, padding=1),
        nn.Dropout(0.1),
        nn.Conv2d(in_channels, out_channels, kernel_size=1))


class FastSCNN(HipModel):
    """(TSS/models/fastscnn.py:15-64)"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.downsample = FusedSequential(
            Conv2dBlock(in_channels, 32, kernel_size=3, padding=1, stride=2),
            DSConv2dBlock(32, 48, kernel_size=3, padding=1, stride=2),
            DSConv2dBlock(48, 64, kernel_size=3, padding=1, stride=2))
        self.features = FusedSequential(
            BottleneckModule(64, 64, expansion=6, repeats=3, stride=2),
            BottleneckModule(64, 96, expansion=6, repeats=3, stride=2),
            BottleneckModule(96, 128, expansion=6, repeats=3, stride=1),
            PyramidPoolingModule(128, 128))
        self.fusion = FeatureFusionModule((128, 64), 128, scale_factor=4)
        self.classifier = Classifier(128, out_channels)

    logit_scale = 8   # the head's F.interpolate(scale_factor=8) (TSS/models/fastscnn.py:63-64)

    def forward_lowres(self, input):
        """Everything up to (not including) the final x8 upsample: (B, classes, H/8, W/8) logits.
        engine.Trainer feeds this to the fused upsample + cross-entropy operator."""
        downsample = self.downsample(self.image_in(input))
        # `downsample` has two consumers (the feature extractor and the fusion module's high-resolution layer): the gradient of the
        # latter is added in the epilogue of the former's first backward-data kernel, not by an elementwise launch (ops.fork_two)
        hooked = has_hooks(self.features) or has_hooks(self.fusion)
        da, db, fork = (downsample, downsample, None) if hooked else ops.fork_two(downsample)
        try:
            features = self.features(da)
            # (the fusion module's high-resolution 1x1 layer only needs `downsample` and could run on a side stream under the feature
            # extractor, as ContextNet's context branch does: measured, 5.58 vs 5.45 ms per step -- it competes with, rather than hides
            # under, the 1/16-resolution kernels; not done)
            fusion = self.fusion(features, db)
        finally:
            if fork is not None:
                ops.drop_fork(da, db)
        return self.classifier(fusion)

    def forward(self, input):
        return self.logits_out(ops.upsample_logits(self.forward_lowres(input), scale_factor=self.logit_scale), input)
