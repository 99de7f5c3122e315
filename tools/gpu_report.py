#!/usr/bin/env python
"""First-contact report (GPU box): run every golden block case through the HIP path and print error tables.
Not a test: never asserts, catches per-case exceptions, so one faulty kernel does not hide the others."""
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.recipe import formula_state            # noqa: E402
from tests import cases                            # noqa: E402

dev = torch.device('cuda:0')


def run_case(name, mode, golden, dtype=torch.float32):
    m = cases.product_block(name)
    m.load_state_dict(formula_state(m), strict=True)
    cases.zero_dropout(m)
    m.train(mode == 'train')
    m.to(dev)
    if dtype != torch.float32:
        import torch_semantic_segmentation_amd as tssa
        tssa.set_compute_dtype(m, dtype)
    xs = []
    for x in cases.block_inputs(name):
        x = x.to(dev)
        if x.shape[1] % 8 == 0:
            x = x.to(dtype)
        xs.append(x.requires_grad_(x.shape[1] % 8 == 0))
    out = m(*xs)
    cot = cases.block_cotangent(out.shape).to(dev).to(out.dtype)
    out.backward(cot)
    torch.cuda.synchronize()
    rows = [('out', cases.rel_err(out.detach().float().cpu().numpy(), golden[name + '/out']))]
    for i, x in enumerate(xs):
        if x.grad is not None:
            rows.append(('dx%d' % i, cases.rel_err(x.grad.float().cpu().numpy(), golden['%s/dx%d' % (name, i)])))
    for pname, p in m.named_parameters():
        g = golden['%s/dw.%s' % (name, pname)]
        if p.grad is None:
            rows.append(('dw.' + pname, float('nan')))
        else:
            rows.append(('dw.' + pname, cases.rel_err(p.grad.cpu().numpy(), g)))
    if mode == 'train':
        for bname, b in m.named_buffers():
            if bname.endswith(('running_mean', 'running_var')):
                rows.append(('buf.' + bname, cases.rel_err(b.cpu().numpy(), golden['%s/buf.%s' % (name, bname)])))
    return rows


def main():
    only = sys.argv[1:]
    gdir = os.path.join(ROOT, 'tests', 'golden')
    for dtype in (torch.float32, torch.bfloat16):
        for mode in ('eval', 'train'):
            golden = cases.load_npz(os.path.join(gdir, 'blocks_%s.npz' % mode))
            for name in sorted(cases.BLOCK_SHAPES):
                if only and name not in only:
                    continue
                t0 = time.time()
                try:
                    rows = run_case(name, mode, golden, dtype)
                    worst = max(rows, key=lambda r: (r[1] if r[1] == r[1] else 1e9))
                    print('%-5s %-5s %-18s worst %-28s %.3e   [%s] %.2fs' % (
                        str(dtype)[6:], mode, name, worst[0], worst[1],
                        ' '.join('%s=%.1e' % (k.replace('running_', 'r'), v) for k, v in rows
                                 if v > (2e-3 if dtype == torch.float32 else 5e-2) or v != v), time.time() - t0))
                except Exception:
                    print('%-5s %-5s %-18s EXCEPTION' % (str(dtype)[6:], mode, name))
                    traceback.print_exc(limit=6)
                sys.stdout.flush()


if __name__ == '__main__':
    main()
