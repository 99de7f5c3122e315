"""Diagnostic (GPU box): per-parameter gradient distance from the f64 reference of the whole frozen ESNet / LedNet -- HIP f32 vs torch f32,
in backward order (last layers first)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import torch_semantic_segmentation_amd as tssa
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
torch.nn.CrossEntropyLoss(ignore_index=255)(o64(x.double()), y).backward()
torch.nn.CrossEntropyLoss(ignore_index=255)(o32(x), y).backward()
tssa.CrossEntropyLoss(ignore_index=255)(m(x.to('cuda:0')), y.to('cuda:0')).backward()
p64, p32, ph = dict(o64.named_parameters()), dict(o32.named_parameters()), dict(m.named_parameters())
for n in reversed(list(p64)):
    r = p64[n].grad
    e = lambda t: float((t.double().cpu() - r).norm() / r.norm().clamp_min(1e-30))
    a, b = e(p32[n].grad), e(ph[n].grad)
    print('%-44s %-18s torch32 %.2e  hip %.2e %s' % (n, tuple(r.shape), a, b, '  <<<' if b > 3 * a + 1e-5 else ''))
