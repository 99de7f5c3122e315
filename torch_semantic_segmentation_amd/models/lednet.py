"""LEDNet's split-shuffle-non-bottleneck unit on the MI355X HIP path: `SSnbtBlock`,
`FactorizedConvBlock`, `ConvBlock` and `channel_shuffle` with the constructor arguments, module tree and state_dict keys of
TSS/models/lednet.py:95-124,147-188.  The factorized 1x3 / 3x1 dilated convolutions run through tss_conv1d3_* (the
implicit-GEMM kernel of the dense 3x3 with three of its nine taps), the channel split is a pair of pitch-addressed views of
the NHWC buffer (no copy), the two branches' BatchNorm is applied while they are written side by side into the concat
buffer, residual + ReLU is the join kernel, and the shuffle is a permutation inside every pixel row (tss_channel_shuffle).

Not covered: nn.Dropout2d with p > 0 in training mode (raises), the encoder's DownsamplingBlock (max-pool + concat + BatchNorm)
and the APN decoder -- the full LEDNet is SURVEY section 8f "next", not this round's scope.
"""
from torch import nn

from .. import ops
from ._fused import FusedSequential, run

__all__ = ['SSnbtBlock', 'FactorizedConvBlock', 'ConvBlock', 'channel_shuffle']


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
        half = x.shape[1] // 2
        if half % 8:
            raise NotImplementedError('HIP path: SSnbtBlock needs a multiple of 16 channels')
        left = run(self.left, x[:, :half])            # torch.chunk(input, 2, 1): two views of the same NHWC rows
        right = run(self.right, x[:, half:])
        y = ops.concat_joined([left, right], relu=False)
        if self.training and self.dropout.p > 0:
            raise NotImplementedError('HIP path: nn.Dropout2d with p > 0 in training mode is not implemented')
        y = ops.join(y, x, relu=True)                 # activation(input + x)
        return channel_shuffle(y, 2)
