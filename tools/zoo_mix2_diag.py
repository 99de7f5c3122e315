"""Diagnostic (GPU box): ESNet layers 1-3 on HIP (k=3) or on torch f32 (k=0), the rest in torch f32: what differs in the logits and in
the cross-entropy gradient."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import torch_semantic_segmentation_amd as tssa
from torch_semantic_segmentation_amd import ops
from tests import cases
from oracle.recipe import synthetic_batch
g = cases.load_npz('tests/golden/zoo_frozen.npz')
name = 'es_net'
def build(dt):
    torch.manual_seed(0)
    o = cases.oracle_zoo(name); cases.zero_all_dropout(o); cases.load_fixture_buffers(o, g, name)
    return o.to(dt).eval()
o32, o64 = build(torch.float32), build(torch.float64)
m = cases.product_zoo(name); m.load_state_dict(o32.state_dict(), strict=True); cases.zero_all_dropout(m)
m.to('cuda:0').eval(); tssa.set_compute_dtype(m, torch.float32)
x, y = synthetic_batch(2, 64, 128)
ce = torch.nn.CrossEntropyLoss(ignore_index=255)
def tail(model, t, k):
    outs = []
    for c in list(model.children())[k:]:
        t = c(t); outs.append(t)
    lg = outs[-1]; lg.retain_grad()
    ce(lg, y).backward()
    return outs
r = tail(o64, x.double(), 0)
res = {}
for k in (0, 3):
    with torch.no_grad():
        t = x.to('cuda:0')
        for c in list(m.children())[:k]: t = ops.materialize(c(t))
        t = t.float().cpu().contiguous()
    res[k] = tail(o32, t, k)
for k in (0, 3):
    lg, lr = res[k][-1], r[-1]
    e = (lg.double() - lr).detach()
    ge = (lg.grad.double() - lr.grad)
    print('k=%d logits: rel %.2e max|e| %.2e at %s | per-class mean e / rms e: %s' % (k, float(e.norm() / lr.norm()), float(e.abs().max()),
          np.unravel_index(int(e.abs().argmax()), e.shape), (e.mean((0, 2, 3)) / e.pow(2).mean((0, 2, 3)).sqrt()).numpy().round(2)))
    print('     dlogits: rel %.2e max|e| %.2e (max|g| %.2e) at %s; top-5 |e|: %s' % (float(ge.norm() / lr.grad.norm()), float(ge.abs().max()), float(lr.grad.abs().max()),
          np.unravel_index(int(ge.abs().argmax()), ge.shape), np.sort(ge.abs().flatten().numpy())[-5:]))
    # stage outputs between
    for i, (a, b) in enumerate(zip(res[k], r[k:] if k else r)):
        d = (a.double() - b).detach()
        print('     stage %d: rel %.2e  max|e| %.2e  top-3 %s' % (i + k, float(d.norm() / b.norm()), float(d.abs().max()), np.sort(d.abs().flatten().numpy())[-3:]))
print('labels at the worst pixel etc: ignore count', int((y == 255).sum()))
