"""Diagnostic (GPU box): ESNet's classifier block + cross-entropy on the REAL layer-5 output (f64 model), so the cotangent is the
cancellation-heavy softmax gradient: HIP f32 vs torch f32 distance from f64, plus the conditioning of the sums involved."""
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
with torch.no_grad():
    t = x.double()
    for n, c in list(o64.named_children())[:-1]:
        t = c(t)
feat = t.float()          # the same f32 input for everybody
def run(cls, inp, ce, dev=None):
    xi = inp.clone().requires_grad_(True)
    out = cls(xi)
    if dev: out = ops.materialize(out)
    out.retain_grad()
    loss = ce(out, y.to(inp.device))
    loss.backward()
    return loss, out, xi.grad
r64 = run(o64.classifier, feat.double(), torch.nn.CrossEntropyLoss(ignore_index=255))
r32 = run(o32.classifier, feat, torch.nn.CrossEntropyLoss(ignore_index=255))
rh = run(m.classifier, feat.to('cuda:0'), tssa.CrossEntropyLoss(ignore_index=255), dev=True)
rel = lambda a, b: float((a.detach().double().cpu() - b.detach().double()).norm() / b.detach().double().norm().clamp_min(1e-30))
print('loss', r64[0].item(), r32[0].item(), rh[0].item())
print('logits   torch32 %.2e hip %.2e' % (rel(r32[1], r64[1]), rel(rh[1], r64[1])))
print('dlogits  torch32 %.2e hip %.2e' % (rel(r32[1].grad, r64[1].grad), rel(rh[1].grad, r64[1].grad)))
print('dx       torch32 %.2e hip %.2e' % (rel(r32[2], r64[2]), rel(rh[2], r64[2])))
for (n, p), (_, q), (_, r) in zip(m.classifier.named_parameters(), o32.classifier.named_parameters(), o64.classifier.named_parameters()):
    print('  %-16s torch32 %.2e hip %.2e' % (n, rel(q.grad, r.grad), rel(p.grad, r.grad)))
# conditioning of d(bn.bias) = sum over pixels of dlogits * mask
gl = r64[1].grad * (r64[1] > 0)
print('cond d(bn.bias): sum|t| / |sum t| per class', (gl.abs().sum((0, 2, 3)) / gl.sum((0, 2, 3)).abs()).numpy().round(1))
# HIP on the f64 gradient of the logits rounded to f32 (takes the cross-entropy out): backward of the block alone
out = ops.materialize(m.classifier(feat.to('cuda:0').requires_grad_(True)))
for p in m.classifier.parameters(): p.grad = None
out.backward(r64[1].grad.float().to('cuda:0'))
for (n, p), (_, r) in zip(m.classifier.named_parameters(), o64.classifier.named_parameters()):
    print('  given f64 dlogits: %-16s hip %.2e' % (n, rel(p.grad, r.grad)))
