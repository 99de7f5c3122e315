import os, sys, importlib, numpy as np, torch
sys.path.insert(0, os.getcwd())
import torch_semantic_segmentation_amd as tssa
from torch_semantic_segmentation_amd import ops
from oracle import aspp as OA
L = importlib.import_module('torch_semantic_segmentation_amd.models.lednet')
for (C, d, H, W, train) in ((128, 9, 8, 16, False), (128, 5, 8, 16, False), (128, 9, 8, 16, True), (128, 17, 8, 16, False), (128, 9, 24, 40, False), (128, 2, 8, 16, False)):
    torch.manual_seed(1)
    ref = OA.SSnbt(C, d)
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5); m.weight.data.uniform_(0.6, 1.4); m.bias.data.uniform_(-0.3, 0.3)
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.p = 0.0
    x = torch.randn(2, C, H, W)
    cot = torch.randn(2, C, H, W)
    ref.train(train)
    xr = x.clone().requires_grad_(True)
    out_r = ref(xr); out_r.backward(cot)
    hip = L.SSnbtBlock(C, C, dilation=d)
    hip.load_state_dict(ref.state_dict(), strict=True)
    for m in hip.modules():
        if isinstance(m, (torch.nn.Dropout, torch.nn.Dropout2d)):
            m.p = 0.0
    hip.to('cuda:0').train(train)
    tssa.set_compute_dtype(hip, torch.float32)
    xh = x.to('cuda:0').requires_grad_(True)
    out_h = ops.materialize(hip(xh)); out_h.backward(cot.to('cuda:0'))
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    print('C %d d %d %dx%d train %d: out %.2e dx %.2e' % (C, d, H, W, train, rel(out_h.cpu(), out_r), rel(xh.grad.cpu(), xr.grad)))
    for (n, p), (_, q) in zip(hip.named_parameters(), ref.named_parameters()):
        e = rel(p.grad.cpu(), q.grad)
        if e > 1e-3:
            print('     %-28s %.3e' % (n, e))
