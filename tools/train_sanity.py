import sys, torch
sys.path.insert(0, '/root/repo')
import torch_semantic_segmentation_amd as tssa
from torch_semantic_segmentation_amd import engine as E
import importlib
name = sys.argv[1] if len(sys.argv) > 1 else 'fastscnn'
mod = importlib.import_module('torch_semantic_segmentation_amd.models.' + {'contextnet14': 'contextnet'}.get(name, name))
ctor = {'fastscnn': lambda: mod.FastSCNN(3, 19), 'contextnet14': lambda: mod.contextnet14(3, 19), 'lednet': lambda: mod.lednet(3, 19),
        'esnet': lambda: mod.ESNet(3, 19)}[name]
torch.manual_seed(0)
dev = 'cuda:0'
model = ctor().to(dev)
tssa.set_compute_dtype(model, torch.bfloat16)
opt = E.FlatAdamW(model.parameters(), lr=2e-3, weight_decay=1e-4)
tr = E.Trainer(model, opt, tssa.CrossEntropyLoss(ignore_index=255), use_graph=True)
B, H, W = (8, 512, 1024) if len(sys.argv) > 2 and sys.argv[2] == 'big' else (4, 256, 512)
x = torch.randn(B, 3, H, W, device=dev)
# make the task learnable: label = function of coarse position
yy = (torch.arange(H, device=dev)[:, None] // 32 + torch.arange(W, device=dev)[None, :] // 64) % 19
y = yy[None].repeat(B, 1, 1).clone(); y[:, :8] = 255
losses = []
for i in range(300):
    l = tr.step_async(x, y)
    if i % 50 == 0 or i == 299:
        losses.append(round(l.item(), 4))
print('loss trajectory', losses)
ev = E.create_segmentation_evaluator(model, None, num_classes=19)
m = ev.run([(x, y)])
print('miou on the training batch', round(m['miou'], 4), 'acc', round(m['accuracy'], 4))
assert losses[-1] < 0.5 * losses[0] and all(v == v for v in losses)
