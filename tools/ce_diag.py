import os, sys, torch
sys.path.insert(0, os.getcwd())
import torch_semantic_segmentation_amd as tssa
from oracle.recipe import synthetic_batch
torch.manual_seed(0)
_, y = synthetic_batch(2, 64, 128)
for scale in (1.0, 5.0):
    z = torch.randn(2, 19, 64, 128) * scale
    z64 = z.double().requires_grad_(True)
    l64 = torch.nn.CrossEntropyLoss(ignore_index=255)(z64, y); l64.backward()
    z32 = z.clone().requires_grad_(True)
    l32 = torch.nn.CrossEntropyLoss(ignore_index=255)(z32, y); l32.backward()
    zh = z.to('cuda:0').requires_grad_(True)
    lh = tssa.CrossEntropyLoss(ignore_index=255)(zh, y.to('cuda:0')); lh.backward()
    zc = z.to('cuda:0').contiguous(memory_format=torch.channels_last).requires_grad_(True)
    lc = tssa.CrossEntropyLoss(ignore_index=255)(zc, y.to('cuda:0')); lc.backward()
    rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
    print('scale %g: loss hip %.3e (channels_last %.3e) torch32 %.3e | grad hip %.3e (channels_last %.3e) torch32 %.3e' % (
        scale, abs(lh.item() / l64.item() - 1), abs(lc.item() / l64.item() - 1), abs(l32.item() / l64.item() - 1),
        rel(zh.grad, z64.grad), rel(zc.grad, z64.grad), rel(z32.grad, z64.grad)))
