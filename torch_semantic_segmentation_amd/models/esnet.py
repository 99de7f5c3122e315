"""ESNet's blocks on the MI355X HIP path: `FCUBlock`, `FPCUBlock` and `DownsamplingBlock` with the
constructor arguments, module tree and state_dict keys of TSS/models/esnet.py:47-68,83-166.

  * the factorized 1xK / Kx1 convolutions (K = 3: tss_conv1d3_*; K = 5: the kh x kw tap grid of the generic implicit-GEMM kernel,
    tss_convkxk_*) carry their bias in the kernel epilogue; conv -> ReLU -> conv -> BatchNorm [-> ReLU] is two deferred units;
  * FCUBlock's `activation(input + dropout(conv2(conv1(input))))` ends in one `join` pass that applies conv2's pending BatchNorm,
    adds the skip and clamps;
  * FPCUBlock's three dilated branches (rates 2 / 5 / 9) read one materialised tensor; `sum(branches)` is two `join` passes that
    apply the branches' BatchNorms while adding;
  * nn.Dropout2d in training mode: ops.channel_dropout (a [B, C] mask, tss_scale_rows);
  * DownsamplingBlock: shared with LEDNet (models/lednet.py downsampling_unit).

Not built: `UpsamplingBlock` (nn.ConvTranspose2d) and therefore the whole `ESNet` -- the decoder half of the model; the transposed
gather it needs exists (tss_convkxk_bwd_data with stride 2 IS that layer's forward), the module around it does not.
"""
from torch import nn

from .. import ops
from ._fused import FusedSequential, run
from .lednet import downsampling_unit

__all__ = ['FCUBlock', 'FPCUBlock', 'DownsamplingBlock']


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
