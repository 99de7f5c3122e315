"""ESNet on the MI355X HIP path: `ESNet`, `FCUBlock`, `FPCUBlock`, `DownsamplingBlock` and `UpsamplingBlock`
with the constructor arguments, module tree and state_dict keys of TSS/models/esnet.py:8-166.

  * the factorized 1xK / Kx1 convolutions (K = 3: tss_conv1d3_*; K = 5: the kh x kw tap grid of the generic implicit-GEMM kernel,
    tss_convkxk_*) carry their bias in the kernel epilogue; conv -> ReLU -> conv -> BatchNorm [-> ReLU] is two deferred units;
  * FCUBlock's `activation(input + dropout(conv2(conv1(input))))` ends in one `join` pass that applies conv2's pending BatchNorm,
    adds the skip and clamps;
  * FPCUBlock's three dilated branches (rates 2 / 5 / 9) read one materialised tensor; `sum(branches)` is two `join` passes that
    apply the branches' BatchNorms while adding;
  * nn.Dropout2d in training mode: ops.channel_dropout (a [B, C] mask, tss_scale_rows);
  * DownsamplingBlock: shared with LEDNet (models/lednet.py downsampling_unit);
  * UpsamplingBlock: nn.ConvTranspose2d(3x3, stride 2, padding 1, output_padding 1) is the transposed gather of the generic
    kernel (tss_convkxk_transposed_fwd; backward = the strided convolution and its weight gradient with the roles swapped), followed
    by a stand-alone deferred BatchNorm (ops.batch_norm); the 19-class classifier runs zero-padded to 24 output channels;
  * `ESNet`: the nn.Sequential of the reference (TSS/models/esnet.py:8-44), returning full-resolution logits.
"""
from collections import OrderedDict

import torch
from torch import nn
from torch.nn import functional as F

from .. import ops
from ._fused import FusedSequential, HipModel, run
from .lednet import downsampling_unit, _PadBN

__all__ = ['ESNet', 'FCUBlock', 'FPCUBlock', 'DownsamplingBlock', 'UpsamplingBlock']


class ESNet(HipModel):
    """(TSS/models/esnet.py:8-44): the reference's nn.Sequential, same child names."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.layer1 = FusedSequential(DownsamplingBlock(in_channels, 16), *[FCUBlock(16, 16, 3, dropout_p=0.03) for _ in range(3)])
        self.layer2 = FusedSequential(DownsamplingBlock(16, 64), *[FCUBlock(64, 64, 5, dropout_p=0.03) for _ in range(2)])
        self.layer3 = FusedSequential(DownsamplingBlock(64, 128), *[FPCUBlock(128, 128, [2, 5, 9], dropout_p=0.3) for _ in range(3)])
        self.layer4 = FusedSequential(UpsamplingBlock(128, 64), FCUBlock(64, 64, 5), FCUBlock(64, 64, 5))
        self.layer5 = FusedSequential(UpsamplingBlock(64, 16), *[FCUBlock(16, 16, 3) for _ in range(3)])
        self.classifier = FusedSequential(UpsamplingBlock(16, out_channels))

    logit_scale = 1      # the classifier already works at full resolution: engine.Trainer takes the loss straight from the NHWC logits
                         # (the fused head + loss operator at scale 1: no NCHW copy of the 8 x 19 x 1024 x 2048 logits, no layout copy back)

    def forward_lowres(self, input):
        x = self.image_in(input)
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4, self.layer5, self.classifier):
            x = stage(x)
        return x

    def forward(self, input):
        return self.logits_out(self.forward_lowres(input), input)


class UpsamplingBlock(nn.Sequential):
    """(TSS/models/esnet.py:71-80)"""

    def __init__(self, in_channels, out_channels):
        super().__init__(OrderedDict([
            ('conv', nn.ConvTranspose2d(in_channels, out_channels, kernel_size=3, stride=2, padding=1, output_padding=1)),
            ('bn', nn.BatchNorm2d(out_channels)),
            ('activation', nn.ReLU(inplace=True)),
        ]))
        self.__dict__['_pad'] = _PadBN()

    def forward(self, input):
        conv, bn = self.conv, self.bn
        C = conv.out_channels
        cp = ops.round_up(C, 8)
        w, b = conv.weight, conv.bias
        if cp != C:
            w = F.pad(w, (0, 0, 0, 0, 0, cp - C))
            b = F.pad(b, (0, cp - C)) if b is not None else None
        y = ops.conv_transpose(input, w, b, conv.stride[0])
        if cp == C:
            return ops.materialize(ops.batch_norm(y, bn, relu=True))
        # ragged class count: BatchNorm of the padded width through a hidden module that carries the running statistics (pad channels
        # are 0 before and after: weight rows 0, bias 0, gamma 1, beta 0)
        pad = self.__dict__['_pad']
        sh = pad.enter(bn, cp)
        out = ops.materialize(ops.batch_norm(y, sh, relu=True, gamma=F.pad(bn.weight, (0, cp - C), value=1.0),
                                             beta=F.pad(bn.bias, (0, cp - C))))
        pad.leave(bn)
        return ops.channel_slice(out, C)


class DownsamplingBlock(nn.Module):
    """(TSS/models/esnet.py:47-68)"""
    act_dtype = None

    def __init__(self, in_channels, out_channels):
        super().__init__()
        if out_channels <= in_channels:
            raise ValueError("output channels must be greater than the input channels")
        self.conv = nn.Conv2d(in_channels, out_channels - in_channels, kernel_size=3, padding=1, stride=2)
        self.pool = nn.MaxPool2d(kernel_size=2)
        self.bn = nn.BatchNorm2d(out_channels)
        self.activation = nn.ReLU(inplace=True)

    def unit(self, d):
        return downsampling_unit(ops.materialize(d), self.conv, self.bn, self.act_dtype)

    def forward(self, input):
        return ops.materialize(self.unit(input))


def _factorized(channels, first, second, padding1, padding2, dilation1=1, dilation2=1, relu=True):
    layers = [nn.Conv2d(channels, channels, kernel_size=first, padding=padding1, dilation=dilation1), nn.ReLU(inplace=True),
              nn.Conv2d(channels, channels, kernel_size=second, padding=padding2, dilation=dilation2), nn.BatchNorm2d(channels)]
    if relu:
        layers.append(nn.ReLU(inplace=True))
    return FusedSequential(*layers)


class FCUBlock(nn.Module):
    """(TSS/models/esnet.py:83-123)"""

    def __init__(self, in_channels, out_channels, kernel_size, dropout_p=0.0):
        super().__init__()
        if in_channels != out_channels:
            raise ValueError("input channels must match output channels")
        k, p = kernel_size, kernel_size // 2
        self.conv1 = _factorized(in_channels, (1, k), (k, 1), (0, p), (p, 0))
        self.conv2 = _factorized(in_channels, (1, k), (k, 1), (0, p), (p, 0), relu=False)
        self.activation = nn.ReLU(inplace=True)
        self.dropout = nn.Dropout2d(p=dropout_p)

    def forward(self, input):
        x = ops.to_nhwc(ops.materialize(input))
        if x.shape[1] % 8:
            raise NotImplementedError('HIP path: FCUBlock needs a multiple of 8 channels')
        y = run(self.conv2, run(self.conv1, x))
        if self.training and self.dropout.p > 0:
            y = ops.channel_dropout(ops.materialize(y), self.dropout.p, True)
        return ops.join(y, x, relu=True)              # activation(input + x): conv2's BatchNorm applied in the same pass


class FPCUBlock(nn.Module):
    """(TSS/models/esnet.py:126-166)"""

    def __init__(self, in_channels, out_channels, dilations, dropout_p=0.0):
        super().__init__()
        if in_channels != out_channels:
            raise ValueError("input channels must match output channels")
        self.conv1 = _factorized(in_channels, (3, 1), (1, 3), (1, 0), (0, 1))
        self.conv2 = nn.ModuleList([
            _factorized(in_channels, (3, 1), (1, 3), (d, 0), (0, d), (d, 1), (1, d), relu=False) for d in dilations])
        self.activation = nn.ReLU(inplace=True)
        self.dropout = nn.Dropout2d(p=dropout_p)

    def forward(self, input):
        x = ops.to_nhwc(ops.materialize(input))
        if x.shape[1] % 8:
            raise NotImplementedError('HIP path: FPCUBlock needs a multiple of 8 channels')
        t = ops.materialize(run(self.conv1, x))        # three consumers
        y = None
        for branch in self.conv2:                      # sum([conv(x) for conv in self.conv2]): BatchNorms applied while adding
            b = run(branch, t)
            y = b if y is None else ops.join(y, b)
        if self.training and self.dropout.p > 0:
            y = ops.channel_dropout(ops.materialize(y), self.dropout.p, True)
        return ops.join(y, x, relu=True)
