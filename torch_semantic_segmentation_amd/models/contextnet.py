"""ContextNet (12/14/18) on the MI355X HIP path: public surface, module tree and state_dict keys (314) of
TSS/models/contextnet.py; arithmetic in the HIP kernels behind include/tss_hip.h.
"""
import torch
from torch import nn

from .. import ops
from ._fused import FusedSequential, HipModel, has_hooks, run

__all__ = ['ContextNet', 'contextnet12', 'contextnet14', 'contextnet18']


def contextnet12(in_channels, out_channels):
    return ContextNet(in_channels, out_channels, scale_factor=2)


def contextnet14(in_channels, out_channels):
    return ContextNet(in_channels, out_channels, scale_factor=4)


def contextnet18(in_channels, out_channels):
    return ContextNet(in_channels, out_channels, scale_factor=8)


def ConvBlock(in_channels, out_channels, kernel_size, padding=0, stride=1, use_relu=True):
    """conv -> BN -> [ReLU]  (TSS/models/contextnet.py:168-177)"""
    layers = [nn.Conv2d(in_channels, out_channels, kernel_size, padding=padding, stride=stride, bias=False),
              nn.BatchNorm2d(out_channels)]
    if use_relu:
        layers.append(nn.ReLU(inplace=True))
    return FusedSequential(*layers)


def DWConvBlock(in_channels, out_channels, kernel_size, padding=0, stride=1, dilation=1, use_relu=True):
    """depthwise conv -> BN -> [ReLU]; same ValueError as the reference (TSS/models/contextnet.py:150-165)"""
    if in_channels != out_channels:
        raise ValueError("input and output channels must be the same in depthwise convolution")
    layers = [nn.Conv2d(in_channels, out_channels, kernel_size, padding=padding, stride=stride,
                        dilation=dilation, groups=in_channels, bias=False),
              nn.BatchNorm2d(out_channels)]
    if use_relu:
        layers.append(nn.ReLU(inplace=True))
    return FusedSequential(*layers)


class BottleneckBlock(nn.Module):
    """(TSS/models/contextnet.py:129-147)"""

    def __init__(self, in_channels, out_channels, stride=1, expansion=6):
        super().__init__()
        mid = in_channels * expansion
        self.conv1 = ConvBlock(in_channels, mid, 1)
        self.conv2 = DWConvBlock(mid, mid, 3, padding=1, stride=stride)
        self.conv3 = ConvBlock(mid, out_channels, 1, use_relu=False)

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
        d = run(self.conv2, run(self.conv1, xa))
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


def LinearBottleneck(in_channels, out_channels, num_blocks, expansion=6, stride=1):
    layers = [BottleneckBlock(in_channels, out_channels, stride=stride, expansion=expansion)]
    layers += [BottleneckBlock(out_channels, out_channels, expansion=expansion) for _ in range(1, num_blocks)]
    return FusedSequential(*layers)


class FeatureFusionModule(nn.Module):
    """(TSS/models/contextnet.py:104-126)"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        lowres_channels, highres_channels = in_channels
        self.lowres = FusedSequential(
            DWConvBlock(lowres_channels, lowres_channels, kernel_size=3, padding=4, dilation=4),
            ConvBlock(lowres_channels, out_channels, 1, use_relu=False))
        self.highres = ConvBlock(highres_channels, out_channels, 1, use_relu=False)

    def forward(self, lowres, highres):
        size = tuple(highres.shape[2:])
        first, rest = self.lowres[0], list(self.lowres)[1:]
        fused = None
        if not has_hooks(self.lowres):
            # interpolate -> DWConvBlock(dilation=4) as one operator (csrc/updw.hip): the upsampled context map is never written
            lowres = ops.to_nhwc(ops.materialize(lowres))
            fused = ops.upsample_dw_unit(lowres, size, first)
        if fused is not None:
            d = fused
            for m in rest:
                d = run(m, d)
        else:
            d = run(self.lowres, ops.bilinear(lowres, size=size))
        return ops.join(d, run(self.highres, highres), relu=True)


def Classifier(in_channels, out_channels):
    """(TSS/models/contextnet.py:79-87)"""
    return FusedSequential(
        DWConvBlock(in_channels, in_channels, 3, padding=1),
        ConvBlock(in_channels, in_channels, 1),
        DWConvBlock(in_channels, in_channels, 3, padding=1),
        ConvBlock(in_channels, in_channels, 1),
        nn.Dropout(p=0.1),
        nn.Conv2d(in_channels, out_channels, 1))


class ContextNet(HipModel):
    """(TSS/models/contextnet.py:28-76)"""

    scale_factor: int = 4

    def __init__(self, in_channels, out_channels, scale_factor=4):
        super().__init__()
        self.scale_factor = scale_factor
        self.spatial = FusedSequential(
            ConvBlock(in_channels, 32, 3, padding=1, stride=2),
            DWConvBlock(32, 32, kernel_size=3, padding=1, stride=2),
            ConvBlock(32, 64, 1),
            DWConvBlock(64, 64, kernel_size=3, padding=1, stride=2),
            ConvBlock(64, 128, 1),
            DWConvBlock(128, 128, kernel_size=3, padding=1, stride=1),
            ConvBlock(128, 128, 1))
        self.context = FusedSequential(
            ConvBlock(in_channels, 32, 3, padding=1, stride=2),
            BottleneckBlock(32, 32, expansion=1),
            BottleneckBlock(32, 32, expansion=6),
            LinearBottleneck(32, 48, 3, stride=2),
            LinearBottleneck(48, 64, 3, stride=2),
            LinearBottleneck(64, 96, 2),
            LinearBottleneck(96, 128, 2),
            ConvBlock(128, 128, 3, padding=1))
        self.feature_fusion = FeatureFusionModule((128, 128), 128)
        self.classifier = Classifier(128, out_channels)

    logit_scale = 8   # the head's F.interpolate(scale_factor=8) (TSS/models/contextnet.py:74-76)

    def _branches_can_overlap(self):
        # not with hooks on the branches (user code would run under the side stream), and not with cross-replica BatchNorm: its
        # all-reduces would be issued from two streams onto one communicator, in an order that can differ between the ranks
        if has_hooks(self.context) or has_hooks(self.spatial):
            return False
        # (torch.nn.SyncBatchNorm too: ops._sync_group activates cross-replica statistics for both classes)
        return not any(isinstance(m, (ops.SyncBatchNorm, torch.nn.SyncBatchNorm)) for m in self.modules())

    def forward_lowres(self, input):
        """Everything up to (not including) the final x8 upsample: (B, classes, H/8, W/8) logits."""
        input = self.image_in(input)
        if ops.overlap_branches and input.is_cuda and self._branches_can_overlap():
            # the two branches are independent up to the fusion module (TSS/models/contextnet.py:66-72): the context branch (a chain of
            # small kernels on the 1/4-resolution image) runs on a side stream under the spatial branch's full-resolution kernels;
            # in a captured step the two become parallel branches of the graph, forward and backward
            dev = input.device
            main, side = torch.cuda.current_stream(dev), ops._side_stream(dev)
            side.wait_stream(main)
            input.record_stream(side)
            with ops.overlap_region():
                with torch.cuda.stream(side):
                    context = ops.resize_image(input, scale_factor=1 / self.scale_factor)
                    context = self.context(context)
                spatial = self.spatial(input)
            main.wait_stream(side)
            for t in ops.tensors_of(context):
                t.record_stream(main)      # allocated on the side stream, consumed (and later freed) under the main one
        else:
            spatial = self.spatial(input)
            context = ops.resize_image(input, scale_factor=1 / self.scale_factor)
            context = self.context(context)
        fusion = self.feature_fusion(context, spatial)
        return self.classifier(fusion)

    def forward(self, input):
        return self.logits_out(ops.upsample_logits(self.forward_lowres(input), scale_factor=self.logit_scale), input)
