"""Diagnostic (GPU box): the first k top-level stages of ESNet on the HIP path, the rest (and the whole backward of the rest) in torch f32:
which stage's forward deviation is the one the last layers' gradients are sensitive to."""
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
ce(o64(x.double()), y).backward()
ref = {n: p.grad.clone() for n, p in o64.named_parameters()}
kids32, kidsh = list(o32.children()), list(m.children())
watch = ['classifier.0.bn.bias', 'classifier.0.conv.weight', 'layer5.3.conv2.3.bias', 'layer5.3.conv2.2.weight']
for k in range(len(kids32) + 1):
    for p in o32.parameters(): p.grad = None
    with torch.no_grad():
        t = x.to('cuda:0')
        for c in kidsh[:k]: t = ops.materialize(c(t))
        t = t.float().cpu().contiguous()
    d = float('nan')
    if k:
        with torch.no_grad():
            r = x.double()
            for c in list(o64.children())[:k]: r = c(r)
        d = float((t.double() - r).norm() / r.norm())
    if k < len(kids32):
        for c in kids32[k:]: t = c(t)
        ce(t, y).backward()
        e = ['%s %.2e' % (n.split('.', 1)[0] + '..' + n.rsplit('.', 2)[-2] + '.' + n.rsplit('.', 1)[-1], float((dict(o32.named_parameters())[n].grad.double() - ref[n]).norm() / ref[n].norm())) for n in watch]
        print('hip stages %d (fwd dev %.1e):' % (k, d), '  '.join(e))
