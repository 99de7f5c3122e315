"""Diagnostic (GPU box): per-parameter gradient norms of the whole LedNet / ESNet frozen-BatchNorm fixture, HIP f32 vs the reference's f64."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import torch_semantic_segmentation_amd as tssa
from tests import cases
from oracle.recipe import synthetic_batch
name = sys.argv[1]
g = cases.load_npz('tests/golden/zoo_frozen.npz')
torch.manual_seed(0)
o = cases.oracle_zoo(name)
m = cases.product_zoo(name)
m.load_state_dict(o.state_dict(), strict=True)
cases.zero_all_dropout(m)
cases.load_fixture_buffers(m, g, name)
m.to('cuda:0').eval()
tssa.set_compute_dtype(m, torch.float32)
x, y = synthetic_batch(2, 64, 128)
loss = tssa.CrossEntropyLoss(ignore_index=255)(m(x.to('cuda:0')), y.to('cuda:0')); loss.backward()
n64, n32, e32 = g[name + '/grad_norms64'], g[name + '/grad_norms32'], g[name + '/err_ref32_per_tensor']
for i, (n, p) in enumerate(m.named_parameters()):
    a = p.grad.double().norm().item()
    print('%-34s hip %11.5g  ref64 %11.5g  ref32 %11.5g   |hip/64-1| %.2e  |32/64-1| %.2e  err32 %.2e' % (n, a, n64[i], n32[i], abs(a / n64[i] - 1), abs(n32[i] / n64[i] - 1), e32[i]))
