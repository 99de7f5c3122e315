#!/usr/bin/env python
"""what does a hipMemsetAsync node do differently from a fill kernel inside the captured step?

Captures the FastSCNN train step twice -- gradient buffer cleared by a kernel (the product: folded into tss_cast_weights) and by
hipMemsetAsync (TSS_MEMSET_NODES=1, a MEMSET node) -- dumps both graphs (hipGraphDebugDotPrint), lists every node that is NOT
reachable from the fill / memset node, and compares the flat gradients of two replays on the same batch from the same weights
(lr = 0), bit for bit, within each variant and across them.  One pass, no loops over the divergent variant.
usage: python tools/graph_memset_probe.py [B H W]  ->  gpurun_out/memset_probe.txt (+ the two .dot files)"""
import os
import re
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_semantic_segmentation_amd as tssa  # noqa: E402
from torch_semantic_segmentation_amd import engine as E  # noqa: E402
from torch_semantic_segmentation_amd.models.fastscnn import fastscnn  # noqa: E402

B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (8, 1024, 2048)
out_dir = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out')
os.makedirs(out_dir, exist_ok=True)
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 3, H, W, generator=g).to(dev)
y = torch.randint(0, 19, (B, H, W), generator=g).to(dev)
report = []


def analyse(dot_path, tag):
    if not os.path.exists(dot_path):
        report.append('%s: no dot file (debug_dump unsupported?)' % tag)
        return
    text = open(dot_path).read()
    nodes = dict(re.findall(r'^\s*"?(\w+)"?\s*\[.*?label\s*=\s*"([^"]*)"', text, flags=re.M | re.S))
    edges = re.findall(r'"?(\w+)"?\s*->\s*"?(\w+)"?', text)
    succ = {}
    for a, b in edges:
        succ.setdefault(a, []).append(b)
    kinds = {n: ('MEMSET' if 'MEMSET' in lab.upper() else ('KERNEL' if 'KERNEL' in lab.upper() or 'kernel' in lab else lab[:20])) for n, lab in nodes.items()}
    roots = [n for n in nodes if 'MEMSET' in kinds[n]] or [n for n, lab in nodes.items() if 'cast_weights' in lab]
    report.append('%s: %d nodes, %d edges, %d fill node(s)' % (tag, len(nodes), len(edges), len(roots)))
    indeg = {n: 0 for n in nodes}
    for a, b in edges:
        if b in indeg:
            indeg[b] += 1
    report.append('%s: root nodes (no incoming edge): %s' % (tag, [nodes[n][:60] for n in nodes if indeg[n] == 0][:8]))
    for r in roots[:4]:
        seen, stack = {r}, [r]
        while stack:
            for b in succ.get(stack.pop(), []):
                if b not in seen:
                    seen.add(b); stack.append(b)
        missing = [n for n in nodes if n not in seen]
        report.append('%s: from fill node "%s": %d reachable, %d NOT reachable' % (tag, nodes[r][:50], len(seen), len(missing)))
        for n in missing[:20]:
            report.append('    not ordered after the fill: %s' % nodes[n][:100])
    fan = sorted(((len(v), nodes.get(k, k)[:60]) for k, v in succ.items() if len(v) > 1), reverse=True)[:6]
    report.append('%s: nodes with more than one successor (parallel branches): %s' % (tag, fan))


def run(memset):
    if memset:
        os.environ['TSS_MEMSET_NODES'] = '1'
    else:
        os.environ.pop('TSS_MEMSET_NODES', None)
    torch.manual_seed(1)
    m = fastscnn(3, 19).to(dev)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    tssa.set_compute_dtype(m, torch.bfloat16)
    opt = E.FlatAdamW(m.parameters(), lr=0.0, weight_decay=0.0)
    tr = E.Trainer(m, opt, tssa.CrossEntropyLoss(ignore_index=255), use_graph=True)
    tr.debug_dot = os.path.join(out_dir, 'step_%s.dot' % ('memset' if memset else 'kernel'))
    grads = []
    for _ in range(3):
        loss = tr.step_async(x, y)
        torch.cuda.synchronize()
        grads.append((float(loss), opt.flat_grad.clone()))
    analyse(tr.debug_dot, 'memset' if memset else 'kernel')
    return grads


gk = run(False)
gm = run(True)
for tag, gs in (('kernel fill', gk), ('memset node', gm)):
    same = [torch.equal(gs[0][1], q[1]) for q in gs[1:]]
    report.append('%s: losses %s; replays bit-identical to the first: %s; max |diff| %s' % (
        tag, ['%.6f' % q[0] for q in gs], same, ['%.3e' % (gs[0][1] - q[1]).abs().max().item() for q in gs[1:]]))
report.append('memset vs kernel, first replay: bit-identical %s, max |diff| %.3e, finite %s' % (
    torch.equal(gk[0][1], gm[0][1]), (gk[0][1] - gm[0][1]).abs().max().item(), bool(torch.isfinite(gm[0][1]).all())))
text = '\n'.join(report)
print(text)
open(os.path.join(out_dir, 'memset_probe.txt'), 'w').write(text + '\n')
