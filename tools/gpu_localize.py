#!/usr/bin/env python
"""Localize a whole-model gradient mismatch: compare d(loss)/d(block output) between the HIP path and the CPU oracle."""
import os, sys
import torch
from torch import nn
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nets as O
from oracle.recipe import formula_state, lattice_input, lattice_target
from tests import cases
import torch_semantic_segmentation_amd as tssa

name = sys.argv[1] if len(sys.argv) > 1 else 'fastscnn'
dev = 'cuda:0'
ref = O.build(name); ref.load_state_dict(formula_state(ref)); cases.zero_dropout(ref); ref.train()
hip = cases.product_model(name); hip.load_state_dict(formula_state(hip)); cases.zero_dropout(hip); hip.to(dev).train()
x = lattice_input(2, 3, 64, 128); y = lattice_target(2, 64, 128)
grads = {}
def tap(model, tag):
    hs = []
    for n, m in model.named_children():
        def hook(mod, inp, out, n=n):
            if isinstance(out, torch.Tensor) and out.requires_grad:
                out.register_hook(lambda g, n=n: grads.__setitem__((tag, n), g.detach().float().cpu()))
        hs.append(m.register_forward_hook(hook))
    return hs
tap(ref, 'ref'); tap(hip, 'hip')
lr = nn.CrossEntropyLoss(ignore_index=255)(ref(x), y); lr.backward()
out_h = hip(x.to(dev)); out_h.register_hook(lambda g: grads.__setitem__(('hip', 'logits'), g.detach().float().cpu()))
out_r = ref(x); out_r.register_hook(lambda g: grads.__setitem__(('ref', 'logits'), g.detach().float().cpu()))
nn.CrossEntropyLoss(ignore_index=255)(out_r, y).backward()
lh = tssa.cross_entropy(out_h, y.to(dev), ignore_index=255); lh.backward()
print('loss', lr.item(), lh.item())
for (tag, n), g in sorted(grads.items()):
    if tag == 'ref' and ('hip', n) in grads:
        h = grads[('hip', n)]
        print('%-16s |ref| %.4e  rel L2 err %.3e  max rel %.3e' % (n, g.norm(), (g - h).norm() / g.norm(), (g - h).abs().max() / g.abs().max()))
pr = dict(ref.named_parameters())
worst = []
for n, p in hip.named_parameters():
    q = pr[n].grad
    e = ((p.grad.cpu() - q).norm() / (q.norm() + 1e-12)).item()
    worst.append((e, n, q.norm().item()))
for e, n, qn in sorted(worst, reverse=True)[:25]:
    print('%-44s rel L2 %.3e |ref| %.3e' % (n, e, qn))
