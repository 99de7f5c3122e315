"""Ad-hoc: are two identical training runs (lr > 0) bit-identical step by step?  Prints the first parameters whose gradient differs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_semantic_segmentation_amd as tssa
from torch_semantic_segmentation_amd import engine as E
from torch_semantic_segmentation_amd.models.fastscnn import fastscnn
from oracle.recipe import synthetic_batch
dev = torch.device('cuda:0')
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 128)
graph = len(sys.argv) > 3 and sys.argv[3] == 'graph'
bf16 = not (len(sys.argv) > 4 and sys.argv[4] == 'f32')
x, y = synthetic_batch(2, H, W)
x, y = x.to(dev), y.to(dev)
def run():
    torch.manual_seed(0)
    m = fastscnn(3, 19).to(dev)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
    if bf16: tssa.set_compute_dtype(m, torch.bfloat16)
    opt = E.FlatAdamW(m.parameters(), lr=1e-3, weight_decay=1e-5)
    tr = E.Trainer(m, opt, tssa.CrossEntropyLoss(ignore_index=255), use_graph=graph)
    out = []
    for _ in range(4):
        loss = tr.step_async(x, y); torch.cuda.synchronize()
        out.append((loss.item(), opt.flat_grad.clone(), opt.flat_param.clone()))
    names = []
    for n_, p_ in m.named_parameters(): names += [n_] * p_.numel()
    return out, names
a, names = run(); b, _ = run()
for i, ((la, ga, pa), (lb, gb, pb)) in enumerate(zip(a, b)):
    bad = sorted({names[j] for j in (ga != gb).nonzero().flatten().tolist()})
    print('step', i, 'loss', la, lb, 'grad equal', torch.equal(ga, gb), 'param equal', torch.equal(pa, pb), bad[:10], len(bad))
