"""Module machinery shared by the FastSCNN / ContextNet mirrors.

The reference builds every block as an ``nn.Sequential`` of ``nn.Conv2d`` / ``nn.BatchNorm2d`` / ``nn.ReLU``
leaves (TSS/models/fastscnn.py:164-199, TSS/models/contextnet.py:150-177).  The mirrors keep exactly that
module tree (same child indices => same ``state_dict`` keys, BatchNorm leaves stay ``_BatchNorm`` instances,
every container stays hookable) but run it through :class:`FusedSequential`, which walks its children and
lowers each ``conv -> [bn] -> [relu]`` group to one HIP unit with deferred BatchNorm (ops.conv_unit).

Hooks: a container is always entered through ``nn.Module.__call__`` when it (or anything below it) has hooks
registered, so forward hooks fire with a real, materialised tensor.  A forward (pre-)hook registered directly on a
``Conv2d`` / ``BatchNorm2d`` / ``ReLU`` leaf of a fused group makes that group run un-deferred: every leaf's hooks fire
with the real tensors the reference's leaf would see (conv output, normalised output, activated output), as
``CaptureOutput`` (TSS/nn/utils.py:7-32) needs.  Such hooks observe; a hook that returns a replacement output, and
backward hooks on these leaves, raise instead of being silently ignored.

dtype: ``model.half()`` / ``model.to(torch.float16 | torch.bfloat16)`` (TSS scripts/contextnet/benchmark_contextnet.py:62)
is accepted: the low-precision activation format of the MI355X path is bfloat16, the kernels read f32 views of the
parameters (cast at the boundary), and the logits come back in the dtype of the input.
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
            'a hook is registered on %s, a leaf inside a fused unit that has no un-fused form; register it on the enclosing '
            'block (any container of this model is hookable)' % m.__class__.__name__)


def _fwd_hooked(m):
    return m is not None and bool(m._forward_hooks or m._forward_pre_hooks)


def _no_backward_hooks(m):
    if m._backward_hooks or m._backward_pre_hooks:
        raise NotImplementedError('backward hooks on %s, a leaf inside a fused conv/bn/relu unit, are not supported; '
                                  'register them on the enclosing block' % m.__class__.__name__)


def _pre_hooks(m, x, may_replace):
    for hid, hook in m._forward_pre_hooks.items():
        kw = getattr(m, '_forward_pre_hooks_with_kwargs', {}).get(hid, False)
        r = hook(m, (x,), {}) if kw else hook(m, (x,))
        if r is not None:
            if not may_replace:
                raise NotImplementedError('a forward pre-hook on %s inside a fused unit may observe its input, not replace it'
                                          % m.__class__.__name__)
            r = r[0] if kw else r
            x = r[0] if isinstance(r, tuple) else r
    return x


def _post_hooks(m, x, out):
    for hid, hook in m._forward_hooks.items():
        kw = getattr(m, '_forward_hooks_with_kwargs', {}).get(hid, False)
        r = hook(m, (x,), {}, out) if kw else hook(m, (x,), out)
        if r is not None:
            raise NotImplementedError('a forward hook on %s inside a fused unit may observe its output, not replace it'
                                      % m.__class__.__name__)


def _hooked_group(d, conv, bn, relu, out_dtype):
    """conv -> [bn] -> [relu] with hooks on at least one leaf: the group runs with every intermediate materialised
    (conv output raw, BatchNorm output, ReLU output are real tensors), hooks fire as nn.Module.__call__ would fire them."""
    for m in (conv, bn, relu):
        if m is not None:
            _no_backward_hooks(m)
    x = ops.materialize(d)
    x = _pre_hooks(conv, x, may_replace=True)
    dd = ops.conv_unit(x, conv, bn, False, out_dtype=out_dtype)
    raw = dd.raw
    _post_hooks(conv, x, raw.detach())
    cur = raw
    if bn is not None:
        _pre_hooks(bn, raw.detach(), may_replace=False)
        cur = ops.join(dd, None, False)
        _post_hooks(bn, raw.detach(), cur)
    if relu is not None:
        _pre_hooks(relu, cur, may_replace=False)
        out = ops.join(Deferred(cur), None, True)
        _post_hooks(relu, cur, out)
        cur = out
    return Deferred(cur)


def run(child, d):
    """Run one child on a Deferred/tensor, staying deferred when the child supports it and has no hooks."""
    if hasattr(child, 'unit') and not has_hooks(child):
        return child.unit(d)
    return ops.as_deferred(child(ops.materialize(d)))


class FusedSequential(nn.Sequential):
    """nn.Sequential executed as one chain of HIP units (see module docstring)."""

    act_dtype = None  # set on stems by set_compute_dtype(): dtype of the first activation

    def unit(self, d, upto=None):
        """upto: run only the first `upto` children (the caller finishes the chain itself, e.g. with a fused epilogue)"""
        mods = list(self) if upto is None else list(self)[:upto]
        i, n = 0, len(mods)
        while i < n:
            m = mods[i]
            if isinstance(m, nn.Conv2d):
                bn = mods[i + 1] if i + 1 < n and isinstance(mods[i + 1], _BatchNorm) else None
                j = i + 1 + (bn is not None)
                relu = j < n and isinstance(mods[j], nn.ReLU)
                leaves = (m, bn, mods[j] if relu else None)
                if any(_fwd_hooked(l) for l in leaves):
                    d = _hooked_group(d, m, bn, mods[j] if relu else None, self.act_dtype)
                else:
                    for l in leaves:
                        if l is not None:
                            _leaf_guard(l)
                    d = ops.conv_unit(d, m, bn, relu, out_dtype=self.act_dtype)
                i = j + int(relu)
            elif isinstance(m, nn.ReLU):
                _leaf_guard(m)
                d = Deferred(ops.join(d, None, True))
                i += 1
            elif isinstance(m, nn.Dropout):
                _leaf_guard(m)
                nxt = mods[i + 1] if i + 1 < n else None
                if (m.training and m.p > 0 and isinstance(nxt, nn.Conv2d) and not _fwd_hooked(nxt)
                        and not (i + 2 < n and isinstance(mods[i + 2], _BatchNorm)) and ops.drop_conv_supported(d, nxt, m.p)):
                    # Dropout -> 1x1 conv (the Classifier's tail, TSS/models/fastscnn.py:96-97): the convolution applies the mask on
                    # load, forward and backward; no pass over the activation just for the dropout
                    _leaf_guard(nxt)
                    d = ops.conv_unit(d, nxt, None, False, out_dtype=self.act_dtype, drop_p=m.p)
                    i += 2
                    continue
                if m.training and m.p > 0:
                    if isinstance(d, Deferred) and d.relu and 0 < m.p < 1 and ops.fuse_dropout:
                        d = Deferred(ops.join(d, None, True, dropout_p=m.p))   # BN + ReLU + dropout in one pass
                    else:
                        d = Deferred(ops.dropout(ops.materialize(d), m.p, True))
                i += 1
            elif isinstance(m, nn.UpsamplingBilinear2d):
                _leaf_guard(m)
                # upsample -> depthwise block (FeatureFusionModule.lowres, TSS/models/fastscnn.py:74-76): one operator that
                # interpolates on the fly (csrc/updw.hip) when the next child is such a block and nobody hooks it
                nxt = mods[i + 1] if i + 1 < n else None
                fused = None
                if isinstance(nxt, FusedSequential) and not has_hooks(nxt):
                    x = ops.to_nhwc(ops.materialize(d))
                    d = x
                    fused = ops.upsample_dw_unit(x, ops._out_size(x, m.size, m.scale_factor), nxt)
                if fused is not None:
                    for l in nxt:
                        _leaf_guard(l)
                    d = fused
                    i += 2
                else:
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


class HipModel(nn.Module):
    """Root-module behaviour shared by the model mirrors: dtype casts of the whole model, as the reference's scripts do
    them (`model.to(device).to(dtype)`, TSS scripts/contextnet/benchmark_contextnet.py:62; apex amp O2 casts to fp16,
    scripts/train_fastscnn.py:147).  Casting the parameters to float16 / bfloat16 selects bfloat16 activations (the
    MI355X counterpart of the reference's fp16 path: same bytes, no loss scaling needed); casting back to float32 restores
    float32 activations.  An explicit set_compute_dtype() on f32 parameters (bf16 activations + f32 masters) is kept."""

    _auto_low = False

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        p = next(self.parameters(), None)
        if p is not None:
            if p.dtype in (torch.float16, torch.bfloat16):
                set_compute_dtype(self, torch.bfloat16)
                self._auto_low = True
            elif self._auto_low and p.dtype == torch.float32:
                set_compute_dtype(self, torch.float32)
                self._auto_low = False
        return out

    @staticmethod
    def image_in(input):
        """float16 images are read as float32 (the stem kernels gather f32 or bf16 planes)."""
        return input.float() if input.dtype == torch.float16 else input

    @staticmethod
    def logits_out(logits, input):
        """forward() returns the dtype it was given when that was a 16-bit float (the reference's modules do)."""
        return logits.to(input.dtype) if input.dtype in (torch.float16, torch.bfloat16) and logits.dtype != input.dtype else logits


def set_compute_dtype(model, dtype):
    """Choose the activation dtype of the HIP path (torch.float32: parity; torch.bfloat16: performance).

    Parameters, BatchNorm statistics and parameter gradients stay float32 (the role apex amp O2's master
    weights play in the reference's scripts, scripts/train_fastscnn.py:147); the image may stay float32.
    """
    if dtype not in (torch.float32, torch.bfloat16):
        raise TypeError('compute dtype must be float32 or bfloat16')
    for m in model.modules():
        if isinstance(m, FusedSequential) or 'act_dtype' in type(m).__dict__:
            m.act_dtype = dtype
    model.compute_dtype = dtype
    return model
