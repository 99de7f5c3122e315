"""Diagnostic (GPU box): where the eval forward of the whole ESNet / LedNet leaves the f64 reference -- per top-level stage, HIP f32 vs torch f32."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import torch_semantic_segmentation_amd as tssa
from torch_semantic_segmentation_amd import ops
from tests import cases
from oracle.recipe import synthetic_batch
name = sys.argv[1]
g = cases.load_npz('tests/golden/zoo_frozen.npz')
def build(dt):
    torch.manual_seed(0)
    o = cases.oracle_zoo(name); cases.zero_all_dropout(o); cases.load_fixture_buffers(o, g, name)
    return o.to(dt).eval()
o32, o64 = build(torch.float32), build(torch.float64)
m = cases.product_zoo(name); m.load_state_dict(o32.state_dict(), strict=True); cases.zero_all_dropout(m)
m.to('cuda:0').eval(); tssa.set_compute_dtype(m, torch.float32)
x, y = synthetic_batch(2, 64, 128)
def stages(model, inp, dev=None):
    outs = []
    t = inp
    kids = list(model.named_children())
    if name == 'led_net':
        kids = list(model.encoder.named_children()) + [('decoder', model.decoder)]
    with torch.no_grad():
        for n, c in kids:
            t = c(t)
            t = ops.materialize(t) if dev else t
            outs.append((n, t.detach().double().cpu()))
    return outs
a64, a32, ah = stages(o64, x.double()), stages(o32, x), stages(m, x.to('cuda:0'), dev=True)
for (n, r), (_, t), (_, h) in zip(a64, a32, ah):
    rel = lambda a: float((a - r).norm() / r.norm())
    print('%-14s %-22s torch32 %.2e   hip %.2e' % (n, tuple(r.shape), rel(t), rel(h)))
# ReLU decisions that differ from the f64 run's (outputs are post-ReLU: > 0 <=> unit open)
for (n, r), (_, t), (_, h) in zip(a64, a32, ah):
    print('%-14s open/closed differs from f64 in: torch32 %d   hip %d   of %d' % (n, int(((t > 0) != (r > 0)).sum()), int(((h > 0) != (r > 0)).sum()), r.numel()))
    for tag, a in (('torch32', t), ('hip', h)):
        d = ((a > 0) != (r > 0))
        if d.any(): print('      %s: |f64 value| at the flips' % tag, np.sort(np.maximum(r[d].abs().numpy(), a[d].abs().numpy()))[-5:])
