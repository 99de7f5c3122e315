"""CPU oracle, part 2: the callers of the hot path, restated (TEST INFRASTRUCTURE ONLY).

* ``train_step``  : TSS/engine.py:24-39  (update_fn of create_segmentation_trainer; ignite and
                    apex are absent from this image, so the body is restated, fp32 branch).
* ``DeepSupervision``: TSS/wrappers/deep_supervision_wrapper.py:10-43.
* ``ohem``        : TSS/losses/ohem_loss.py:10-21.
* ``formula_state`` / ``lattice_input`` / ``synthetic_batch``: deterministic, closed-form weights and
  inputs so golden fixtures only need to store outputs (SURVEY.md §8c "golden vectors").
"""
import math
from functools import partial

import torch
from torch import nn
from torch.nn import functional as F


# ----------------------------------------------------------------------------- inputs / weights

def formula_state(module, gain=1.0, salt=0.0):
    """Closed-form, well-conditioned state_dict for any module tree (no RNG, no files).

    conv weight  w[i] = gain * sqrt(3/fan_in) * sin(0.37*i + 1.3*k + salt)   (k = tensor ordinal)
    bn weight    1 + 0.25*sin(0.91*i + k)          bn bias   0.1*cos(0.53*i + k)
    running_mean 0.05*sin(0.71*i + k)              running_var 1 + 0.3*cos(0.29*i + k)
    conv bias    0.1*sin(0.77*i + k)               num_batches_tracked 0
    """
    out = {}
    for k, (name, t) in enumerate(module.state_dict().items()):
        i = torch.arange(t.numel(), dtype=torch.float64)
        if name.endswith('num_batches_tracked'):
            v = torch.zeros((), dtype=torch.int64)
        elif name.endswith('running_mean'):
            v = 0.05 * torch.sin(0.71 * i + k)
        elif name.endswith('running_var'):
            v = 1.0 + 0.3 * torch.cos(0.29 * i + k)
        elif t.dim() == 4:
            fan_in = t.shape[1] * t.shape[2] * t.shape[3]
            v = gain * math.sqrt(3.0 / fan_in) * torch.sin(0.37 * i + 1.3 * k + salt)
        elif name.endswith('weight'):
            v = 1.0 + 0.25 * torch.sin(0.91 * i + k)
        elif t.dim() == 1 and name.endswith('bias') and _is_conv_bias(module, name):
            v = 0.1 * torch.sin(0.77 * i + k)
        else:
            v = 0.1 * torch.cos(0.53 * i + k)
        out[name] = v.reshape(t.shape).to(t.dtype)
    return out


def _is_conv_bias(module, name):
    owner = module.get_submodule(name.rsplit('.', 1)[0]) if '.' in name else module
    return isinstance(owner, nn.Conv2d)


def lattice_input(b, c, h, w, dtype=torch.float32):
    """x[b,c,y,x] = sin(0.11*y + 0.07*x + 0.9*c + 0.5*b) + 0.5*cos(0.013*y*x + c)  (closed form)."""
    bb = torch.arange(b, dtype=torch.float64).view(b, 1, 1, 1)
    cc = torch.arange(c, dtype=torch.float64).view(1, c, 1, 1)
    yy = torch.arange(h, dtype=torch.float64).view(1, 1, h, 1)
    xx = torch.arange(w, dtype=torch.float64).view(1, 1, 1, w)
    v = torch.sin(0.11 * yy + 0.07 * xx + 0.9 * cc + 0.5 * bb) + 0.5 * torch.cos(0.013 * yy * xx + cc)
    return v.to(dtype)


def lattice_target(b, h, w, classes=19, ignore=255):
    """int64 labels 0..classes-1 in a blocky pattern, ~4% set to `ignore` (closed form)."""
    bb = torch.arange(b).view(b, 1, 1)
    yy = torch.arange(h).view(1, h, 1)
    xx = torch.arange(w).view(1, 1, w)
    t = ((yy // 3) * 5 + (xx // 4) * 3 + bb * 7) % classes
    hole = ((yy * 31 + xx * 17 + bb * 5) % 25) == 0
    return torch.where(hole, torch.full_like(t, ignore), t).to(torch.int64)


def synthetic_batch(b, h, w, classes=19, seed=1234, ignore=255, ignore_frac=0.05, in_channels=3):
    """SURVEY.md §8d inputs: x ~ N(0,1) fp32, y uniform over classes with 5% ignore, one generator."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, in_channels, h, w, generator=g, dtype=torch.float32)
    y = torch.randint(0, classes, (b, h, w), generator=g, dtype=torch.int64)
    drop = torch.rand(b, h, w, generator=g) < ignore_frac
    y[drop] = ignore
    return x, y


# ----------------------------------------------------------------------------- callers

def train_step(model, optimizer, loss_fn, x, y):
    """One iteration of the reference trainer, fp32 branch (TSS/engine.py:25-39)."""
    model.train()
    optimizer.zero_grad()
    y_pred = model(x)
    loss = loss_fn(y_pred, y)
    loss.backward()
    optimizer.step()
    return loss.item()


class DeepSupervision(nn.Module):
    """Training: (output, [aux_i(layer_i output)]) via forward hooks; eval: output only."""

    def __init__(self, module, auxiliary_modules):
        super().__init__()
        self.module = module
        self.layers = [layer for layer, _ in auxiliary_modules]
        self.auxiliary = nn.ModuleList([head for _, head in auxiliary_modules])

    def forward(self, x):
        if not self.training:
            return self.module(x)
        aux = [None] * len(self.layers)

        def grab(_m, _inp, out, slot, head):
            aux[slot] = head(out)

        handles = [layer.register_forward_hook(partial(grab, slot=i, head=head))
                   for i, (layer, head) in enumerate(zip(self.layers, self.auxiliary))]
        out = self.module(x)
        for h in handles:
            h.remove()
        return out, aux


def ohem(logits, target, ignore_index=-100, thresh_loss=-math.log(0.7), numel_frac=0.01):
    """Online hard example mining CE (TSS/losses/ohem_loss.py:10-21): ignored pixels count in numel."""
    per_pixel = F.cross_entropy(logits, target, ignore_index=ignore_index, reduction='none').flatten()
    n = int(per_pixel.numel() * numel_frac)
    ordered, _ = torch.sort(per_pixel, descending=True)
    if ordered[n] > thresh_loss:
        return ordered[ordered > thresh_loss].mean()
    return ordered[:n].mean()
