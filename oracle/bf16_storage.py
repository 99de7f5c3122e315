"""CPU oracle, part 3 (TEST INFRASTRUCTURE ONLY): the reference's arithmetic with the STORAGE FORMAT of the bf16 path.

The HIP path with bf16 activations stores every convolution output, every block output and every gradient that
travels between two layers as bfloat16; everything else -- BatchNorm statistics, the normalisation
applied on load, accumulators -- is f32/f64.  Comparing it with a pure f64 run of the reference mixes two effects: the
2^-9 rounding of stored tensors (harmless, random) and the ReLU masks that flip where a rounded pre-activation crosses
zero (0.2-0.5 % of the elements; each flip moves a gradient element by its full magnitude, so weight gradients differ
by 4-8 % in relative L2 although nothing is wrong).  To pin the bf16 KERNELS tightly, the oracle here runs the same
reference modules (oracle/nets.py, which restates TSS/models/fastscnn.py and TSS/models/contextnet.py) in float64 and
rounds exactly the tensors the HIP path stores:

  * forward : the output of every nn.Conv2d, of every pooling / upsampling module and of every block that the product
              materialises (the residual blocks, the fusion modules, the pyramid module, the root) is rounded to bf16;
  * forward : the INPUT of every dense convolution (1x1, 3x3, stem) is rounded to bf16 as well -- the matrix-core kernels
              normalise their input tile and hand it to the MFMA as bf16; depthwise convolutions consume it in f32;
  * backward: the gradient arriving at each of those tensors is rounded to bf16 (what the backward kernels store, and
              what they feed to the MFMA).

The masks of both sides then agree (they are computed from the same rounded pre-activations), and what is left is the
accumulation order and the rounding of MFMA operands: ~1e-3..1e-2 relative L2 instead of 4-8e-2.
"""
import torch
from torch import nn

from . import nets


class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def round_bf16(x):
    return _RoundBF16.apply(x)


# ---- noise model of the storage format.  Which of its two bf16 neighbours a stored value lands on depends on accumulation
# order (a kernel sums its f32 products in another order than this oracle sums its f64 ones), i.e. for the purpose of a bound
# it is a coin that is the more biased the closer the exact value sits to one neighbour.  Dithered rounding draws exactly that
# coin: bf16(x + u * ulp(x)), u uniform in (-1/2, 1/2).  Two oracle runs with independent draws differ by what two CORRECT
# implementations of the same storage format may differ by -- including the ReLU masks that flip where a pre-activation
# rounds across zero -- and tests/test_gpu_lean_vs_oracle.py bounds every kernel by a multiple of that distance, tensor by
# tensor, instead of by a cap read off a previous run.
class _DitherBF16(torch.autograd.Function):
    @staticmethod
    def _dither(x, gen):
        ax = x.abs().clamp_min(torch.finfo(torch.float64).tiny if x.dtype == torch.float64 else 1e-38)
        ulp = torch.exp2(torch.floor(torch.log2(ax)) - 7.0)           # bf16: 8 significant bits
        u = torch.rand(x.shape, generator=gen, dtype=x.dtype) - 0.5
        return (x + u * ulp).to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def forward(ctx, x, gen):
        ctx.gen = gen
        return _DitherBF16._dither(x, gen)

    @staticmethod
    def backward(ctx, g):
        return _DitherBF16._dither(g, ctx.gen), None


from . import aspp as _aspp
from . import zoo as _zoo

_STORED = (nn.Conv2d, nn.ConvTranspose2d, nn.AdaptiveAvgPool2d, nn.UpsamplingBilinear2d, nn.Upsample,
           nets.InvertedResidual, nets.Pyramid, nets.FastFusion, nets.CtxFusion, _aspp.SSnbt,
           _zoo.FactorizedUnit, _zoo.ParallelFactorizedUnit)


def emulate_bf16_storage(module, root=True, dither=None):
    """Register forward hooks on `module` (an oracle module tree) that round the tensors the bf16 HIP path stores.
    `dither`: a torch.Generator -- round with the noise model above (independent draws per tensor) instead of to nearest.
    Returns the hook handles (call .remove() on each to undo)."""
    handles = []
    rnd = round_bf16 if dither is None else (lambda t: _DitherBF16.apply(t, dither))

    def hook(_m, _inp, out):
        return rnd(out) if torch.is_tensor(out) else out

    def pre_hook(_m, inp):
        return tuple(rnd(t) if torch.is_tensor(t) and t.is_floating_point() else t for t in inp)
    for m in module.modules():
        if isinstance(m, _STORED) or (root and m is module):
            handles.append(m.register_forward_hook(hook))
        if isinstance(m, nn.Conv2d) and m.groups == 1:
            handles.append(m.register_forward_pre_hook(pre_hook))
    return handles
