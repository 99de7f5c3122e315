"""LEDNet's split-shuffle-non-bottleneck unit on the MI355X HIP path: `SSnbtBlock`,
`FactorizedConvBlock`, `ConvBlock` and `channel_shuffle` with the constructor arguments, module tree and state_dict keys of
TSS/models/lednet.py:95-124,147-188.  The factorized 1x3 / 3x1 dilated convolutions run through tss_conv1d3_* (the
implicit-GEMM kernel of the dense 3x3 with three of its nine taps), the channel split is a pair of pitch-addressed views of
the NHWC buffer (no copy), the two branches' BatchNorm is applied while they are written side by side into the concat
buffer, residual + ReLU is the join kernel, and the shuffle is a permutation inside every pixel row (tss_channel_shuffle).

The whole model (TSS/models/lednet.py:13-92,126-144) on top of that unit:
  * `DownsamplingBlock`: the strided 3x3 convolution (stem kernel on the image, generic implicit-GEMM kernel with its transposed
    gather in backward otherwise) and the 2x2 max-pool are written side by side into one NHWC buffer (tss_pool_concat_*, the
    convolution's bias added there), whose BatchNorm is a deferred unit on a materialised tensor (ops.batch_norm);
  * `APNModule`: 3x3 / 5x5 / 7x7 stride-2 ConvBlocks on the kh x kw tap grid of the generic kernel (tss_convkxk_*); the five
    1x1 `level` layers run with their `out_channels` zero-padded to a multiple of 8 (19 classes -> 24: pad channels are exactly
    0 through BatchNorm, ReLU, upsampling, sums and products, and are sliced away before the x8 head), the pyramid sums are `join`
    passes, `x * level4 + level5(pooled)` is tss_mul_addrows_*;
  * nn.Dropout2d in training mode: a [B, C] mask drawn with torch's generator, applied by tss_scale_rows (its own backward).
Module tree, constructor arguments and state_dict keys are the reference's.
"""
from collections import OrderedDict

import torch
from torch import nn
from torch.nn import functional as F

from .. import ops
from ._fused import FusedSequential, HipModel, run

__all__ = ['LedNet', 'lednet', 'APNModule', 'DownsamplingBlock', 'SSnbtBlock', 'FactorizedConvBlock', 'ConvBlock', 'channel_shuffle']


def lednet(in_channels, out_channels):
    """(TSS/models/lednet.py:13-14)"""
    return LedNet(in_channels, out_channels)


class LedNet(HipModel):
    """(TSS/models/lednet.py:17-55)"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.encoder = FusedSequential(OrderedDict([
            ('layer1', FusedSequential(
                DownsamplingBlock(in_channels, 32),
                SSnbtBlock(32, 32, dropout_p=0.03),
                SSnbtBlock(32, 32, dropout_p=0.03),
                SSnbtBlock(32, 32, dropout_p=0.03))),
            ('layer2', FusedSequential(
                DownsamplingBlock(32, 64),
                SSnbtBlock(64, 64, dropout_p=0.03),
                SSnbtBlock(64, 64, dropout_p=0.03))),
            ('layer3', FusedSequential(
                DownsamplingBlock(64, 128),
                SSnbtBlock(128, 128),
                SSnbtBlock(128, 128, dilation=2, dropout_p=0.3),
                SSnbtBlock(128, 128, dilation=5, dropout_p=0.3),
                SSnbtBlock(128, 128, dilation=9, dropout_p=0.3),
                SSnbtBlock(128, 128, dilation=2, dropout_p=0.3),
                SSnbtBlock(128, 128, dilation=5, dropout_p=0.3),
                SSnbtBlock(128, 128, dilation=9, dropout_p=0.3),
                SSnbtBlock(128, 128, dilation=17, dropout_p=0.3))),
        ]))
        self.decoder = APNModule(128, out_channels)

    logit_scale = 8

    def forward_lowres(self, input):
        """Everything up to the final x8 upsample (engine.Trainer fuses that with the loss)."""
        x = self.encoder(self.image_in(input))
        return self.decoder(x)

    def forward(self, input):
        return self.logits_out(ops.upsample_logits(self.forward_lowres(input), scale_factor=self.logit_scale), input)


class _PadBN:
    """The BatchNorm of a layer that runs with zero-padded output channels: a hidden nn.BatchNorm2d of the padded width that
    carries the running statistics through the kernels; the real module's buffers are filled from it after every training step.
    Not a submodule (no state_dict keys, no parameters)."""

    def __init__(self):
        self.shim = None

    def enter(self, bn, cp):
        C = bn.num_features
        dev = bn.weight.device
        if self.shim is None or self.shim.num_features != cp or self.shim.running_mean.device != dev:
            self.shim = nn.BatchNorm2d(cp, eps=bn.eps, momentum=bn.momentum, affine=False,
                                       track_running_stats=bn.track_running_stats).to(dev)
        sh = self.shim
        sh.eps, sh.momentum = bn.eps, bn.momentum
        sh.train(bn.training)
        if bn.track_running_stats and bn.running_mean is not None:
            with torch.no_grad():
                sh.running_mean[:C].copy_(bn.running_mean)
                sh.running_var[:C].copy_(bn.running_var)
                sh.num_batches_tracked.copy_(bn.num_batches_tracked)
        return sh

    def leave(self, bn):
        if bn.training and bn.track_running_stats and bn.running_mean is not None:
            C, sh = bn.num_features, self.shim
            with torch.no_grad():
                bn.running_mean.copy_(sh.running_mean[:C])
                bn.running_var.copy_(sh.running_var[:C])
                bn.num_batches_tracked.copy_(sh.num_batches_tracked)


def _padded_block(x, block, cp, pad):
    """ConvBlock(in, C, 1) with C padded to cp output channels (zero weight rows, gamma 1, beta 0): Deferred of cp channels."""
    conv, bn = block[0], block[1]
    C = conv.out_channels
    if C == cp:
        return run(block, x)
    w = F.pad(conv.weight, (0, 0, 0, 0, 0, 0, 0, cp - C))
    gamma, beta = F.pad(bn.weight, (0, cp - C), value=1.0), F.pad(bn.bias, (0, cp - C))
    sh = pad.enter(bn, cp)
    out = ops.conv_unit(x, conv, sh, True, weight=w, gamma=gamma, beta=beta, cout=cp)
    pad.leave(bn)
    return out


class APNModule(nn.Module):
    """(TSS/models/lednet.py:58-92).  Returns the logits with `out_channels` channels (a pitch-addressed view of the padded buffer)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = ConvBlock(in_channels, in_channels, 3, 1, stride=2)
        self.conv2 = ConvBlock(in_channels, in_channels, 5, 2, stride=2)
        self.conv3 = ConvBlock(in_channels, in_channels, 7, 3, stride=2)
        self.level1 = ConvBlock(in_channels, out_channels, 1)
        self.level2 = ConvBlock(in_channels, out_channels, 1)
        self.level3 = ConvBlock(in_channels, out_channels, 1)
        self.level4 = ConvBlock(in_channels, out_channels, 1)
        self.level5 = ConvBlock(in_channels, out_channels, 1)
        self.__dict__['_pads'] = [_PadBN() for _ in range(5)]

    def forward(self, input):
        x = ops.to_nhwc(ops.materialize(input))
        C = self.level1[0].out_channels
        cp = ops.round_up(C, 8)
        pads = self.__dict__['_pads']
        b3 = run(self.conv1, x)                       # 1/2, Deferred (BatchNorm + ReLU pending)
        b3 = ops.materialize(b3)                      # two consumers: conv2 and level3
        b2 = ops.materialize(run(self.conv2, b3))
        b1 = run(self.conv3, b2)
        l1 = ops.materialize(_padded_block(b1, self.level1, cp, pads[0]))
        l2 = ops.materialize(_padded_block(b2, self.level2, cp, pads[1]))
        l3 = ops.materialize(_padded_block(b3, self.level3, cp, pads[2]))
        up = ops.bilinear(l1, scale_factor=2)
        up = ops.bilinear(ops.join(l2, up), scale_factor=2)
        up = ops.bilinear(ops.join(l3, up), scale_factor=2)
        l4 = _padded_block(x, self.level4, cp, pads[3])
        pooled = ops.adaptive_avg_pool(x, 1)
        l5 = _padded_block(pooled, self.level5, cp, pads[4])
        out = ops.mul_addrows(up, l4, l5)
        return ops.channel_slice(out, C)


class DownsamplingBlock(nn.Module):
    """(TSS/models/lednet.py:126-144)"""
    act_dtype = None      # set by set_compute_dtype(): dtype of the first activation when the input is the image

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels - in_channels, kernel_size=3, padding=1, stride=2)
        self.pool = nn.MaxPool2d(kernel_size=2)
        self.bn = nn.BatchNorm2d(out_channels)
        self.relu = nn.ReLU(inplace=True)

    def unit(self, d):
        x = ops.materialize(d)
        return downsampling_unit(x, self.conv, self.bn, self.act_dtype)

    def forward(self, input):
        return ops.materialize(self.unit(input))


def downsampling_unit(x, conv, bn, act_dtype=None):
    """relu(bn(cat([conv(x), max_pool2d(x, 2)]))) as a deferred unit; shared with ESNet's DownsamplingBlock (TSS/models/esnet.py:47-68)."""
    n1 = conv.out_channels
    cp = ops.round_up(n1, 8)
    w = conv.weight if cp == n1 else F.pad(conv.weight, (0, 0, 0, 0, 0, 0, 0, cp - n1))
    is_image = x.shape[1] % 8 != 0
    if is_image or conv.bias is None:      # the stem kernel has no bias input: the bias is added while the halves are written side by side
        y1 = ops.conv_unit(x, conv, None, False, out_dtype=act_dtype, weight=w, bias=None, cout=cp)
        z = ops.pool_concat(y1.raw, conv.bias, x, n1)
    else:                                   # bias in the convolution's epilogue: conv(x) + bias is rounded once, as the reference's tensor is
        b = conv.bias if cp == n1 else F.pad(conv.bias, (0, cp - n1))
        y1 = ops.conv_unit(x, conv, None, False, out_dtype=act_dtype, weight=w, bias=b, cout=cp)
        z = ops.pool_concat(y1.raw, None, x, n1)
    return ops.batch_norm(z, bn, relu=True)


def ConvBlock(in_channels, out_channels, kernel_size, padding=0, stride=1):
    """(TSS/models/lednet.py:147-154)"""
    return FusedSequential(nn.Conv2d(in_channels, out_channels, kernel_size, padding=padding, stride=stride, bias=False),
                           nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))


def FactorizedConvBlock(in_channels, out_channels, dilation=1, use_relu=True):
    """1x3 conv -> ReLU -> 3x1 conv -> BatchNorm -> [ReLU]  (TSS/models/lednet.py:157-180)"""
    if in_channels != out_channels:
        raise ValueError("input and output channels must match")
    layers = [nn.Conv2d(in_channels, in_channels, kernel_size=(1, 3), padding=(0, dilation), dilation=(1, dilation), bias=False),
              nn.ReLU(inplace=True),
              nn.Conv2d(in_channels, in_channels, kernel_size=(3, 1), padding=(dilation, 0), dilation=(dilation, 1), bias=False),
              nn.BatchNorm2d(in_channels)]
    if use_relu:
        layers += [nn.ReLU(inplace=True)]
    return FusedSequential(*layers)


def channel_shuffle(x, groups):
    """(TSS/models/lednet.py:183-188)"""
    return ops.channel_shuffle(x, groups)


class SSnbtBlock(nn.Module):
    """(TSS/models/lednet.py:95-124)"""

    def __init__(self, in_channels, out_channels, dilation=1, dropout_p=0.0):
        super().__init__()
        if in_channels != out_channels:
            raise ValueError("input and output channels must match")
        channels = in_channels // 2
        self.left = FusedSequential(FactorizedConvBlock(channels, channels),
                                    FactorizedConvBlock(channels, channels, dilation, use_relu=False))
        self.right = FusedSequential(FactorizedConvBlock(channels, channels),
                                     FactorizedConvBlock(channels, channels, dilation, use_relu=False))
        self.activation = nn.ReLU(inplace=True)
        self.dropout = nn.Dropout2d(p=dropout_p)

    def forward(self, input):
        x = ops.to_nhwc(ops.materialize(input))
        if x.shape[1] % 16:
            raise NotImplementedError('HIP path: SSnbtBlock needs a multiple of 16 channels')
        xl, xr, x = ops.split_fork(x)                 # torch.chunk(input, 2, 1) + the skip: three views of the same NHWC rows
        left = run(self.left, xl)
        right = run(self.right, xr)
        # cat -> Dropout2d -> activation(input + x) -> channel_shuffle(x, 2): one pass (csrc/ssnbt.hip)
        return ops.ssnbt_tail(left, right, x, self.dropout.p, self.training)
