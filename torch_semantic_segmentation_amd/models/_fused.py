"""Module machinery shared by the FastSCNN / ContextNet mirrors.

The reference builds every block as an ``nn.Sequential`` of ``nn.Conv2d`` / ``nn.BatchNorm2d`` / ``nn.ReLU``
leaves (TSS/models/fastscnn.py:164-199, TSS/models/contextnet.py:150-177).  The mirrors keep exactly that
module tree (same child indices => same ``state_dict`` keys, BatchNorm leaves stay ``_BatchNorm`` instances,
every container stays hookable) but run it through :class:`FusedSequential`, which walks its children and
lowers each ``conv -> [bn] -> [relu]`` group to one HIP unit with deferred BatchNorm (ops.conv_unit).

Hooks: a container is always entered through ``nn.Module.__call__`` when it (or anything below it) has hooks
registered, so forward hooks fire with a real, materialised tensor.  Leaf modules inside a fused group are not
called individually; a hook registered directly on such a leaf raises instead of being silently skipped.
"""
import torch
from torch import nn
from torch.nn.modules.batchnorm import _BatchNorm
from torch.nn.modules import module as _module_mod

from .. import ops

Deferred = ops.Deferred


def _own_hooks(m):
    return bool(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks
                or getattr(m, '_forward_hooks_with_kwargs', None))


def _global_hooks():
    g = _module_mod
    return bool(g._global_forward_hooks or g._global_forward_pre_hooks or g._global_backward_hooks
                or getattr(g, '_global_backward_pre_hooks', None))


def has_hooks(m):
    """True if calling `m` through the fused path would skip a registered hook."""
    if _global_hooks():
        return True
    return any(_own_hooks(s) for s in m.modules())


def _leaf_guard(m):
    if _own_hooks(m):
        raise NotImplementedError(
            'a hook is registered on %s, a leaf inside a fused conv/bn/relu unit; register it on the enclosing '
            'block (any container of this model is hookable)' % m.__class__.__name__)


def run(child, d):
    """Run one child on a Deferred/tensor, staying deferred when the child supports it and has no hooks."""
    if hasattr(child, 'unit') and not has_hooks(child):
        return child.unit(d)
    return ops.as_deferred(child(ops.materialize(d)))


class FusedSequential(nn.Sequential):
    """nn.Sequential executed as one chain of HIP units (see module docstring)."""

    act_dtype = None  # set on stems by set_compute_dtype(): dtype of the first activation

    def unit(self, d):
        mods = list(self)
        i, n = 0, len(mods)
        while i < n:
            m = mods[i]
            if isinstance(m, nn.Conv2d):
                _leaf_guard(m)
                bn = mods[i + 1] if i + 1 < n and isinstance(mods[i + 1], _BatchNorm) else None
                j = i + 1 + (bn is not None)
                relu = j < n and isinstance(mods[j], nn.ReLU)
                if bn is not None:
                    _leaf_guard(bn)
                if relu:
                    _leaf_guard(mods[j])
                d = ops.conv_unit(d, m, bn, relu, out_dtype=self.act_dtype)
                i = j + int(relu)
            elif isinstance(m, nn.ReLU):
                _leaf_guard(m)
                d = Deferred(ops.join(d, None, True))
                i += 1
            elif isinstance(m, nn.Dropout):
                _leaf_guard(m)
                if m.training and m.p > 0:
                    if isinstance(d, Deferred) and d.relu and 0 < m.p < 1 and ops.fuse_dropout:
                        d = Deferred(ops.join(d, None, True, dropout_p=m.p))   # BN + ReLU + dropout in one pass
                    else:
                        d = Deferred(ops.dropout(ops.materialize(d), m.p, True))
                i += 1
            elif isinstance(m, nn.UpsamplingBilinear2d):
                _leaf_guard(m)
                d = Deferred(ops.bilinear(d, size=m.size, scale_factor=m.scale_factor))
                i += 1
            elif isinstance(m, nn.AdaptiveAvgPool2d):
                _leaf_guard(m)
                size = m.output_size
                if not isinstance(size, int):
                    if size[0] != size[1]:
                        raise NotImplementedError('HIP path: square adaptive pooling only')
                    size = size[0]
                d = Deferred(ops.adaptive_avg_pool(d, size))
                i += 1
            else:
                d = run(m, d)
                i += 1
        return d

    def forward(self, input):
        return ops.materialize(self.unit(input))


def set_compute_dtype(model, dtype):
    """Choose the activation dtype of the HIP path (torch.float32: parity; torch.bfloat16: performance).

    Parameters, BatchNorm statistics and parameter gradients stay float32 (the role apex amp O2's master
    weights play in the reference's scripts, scripts/train_fastscnn.py:147); the image may stay float32.
    """
    if dtype not in (torch.float32, torch.bfloat16):
        raise TypeError('compute dtype must be float32 or bfloat16')
    for m in model.modules():
        if isinstance(m, FusedSequential):
            m.act_dtype = dtype
    model.compute_dtype = dtype
    return model
