"""Diagnostic (GPU box): inside ESNet's layer3, sub-block by sub-block: size and COHERENCE (per-channel mean of the signed error over its
rms) of the f32 forward deviation from f64, HIP vs torch."""
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
    r = x.double()
    for c in list(o64.children())[:2]: r = c(r)
    feat = r.float()
    r, t, h = feat.double(), feat, feat.to('cuda:0')
    for i, (c64, c32, ch) in enumerate(zip(o64.layer3.children(), o32.layer3.children(), m.layer3.children())):
        r, t, h = c64(r), c32(t), ops.materialize(ch(h))
        for tag, a in (('torch32', t), ('hip', h)):
            e = a.double().cpu() - r
            rms = e.pow(2).mean((0, 2, 3)).sqrt()
            coh = (e.mean((0, 2, 3)).abs() / rms.clamp_min(1e-30))
            print('layer3.%d %-8s rel %.2e  max|e| %.2e (max|ref| %.2e)  coherence mean %.3f max %.3f   ch of max %d' % (
                i, tag, float(e.norm() / r.norm()), float(e.abs().max()), float(r.abs().max()), float(coh.mean()), float(coh.max()), int(coh.argmax())))
