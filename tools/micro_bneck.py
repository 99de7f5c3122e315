#!/usr/bin/env python
"""Microbenchmark (GPU box): the eval-mode inverted residual as one kernel (csrc/bneck.hip) at BASELINE config 5's block shapes
(1 x 3 x 2048 x 4096 image), against the layer-by-layer eval path; with a TSS_TIMING=1 build also the cycles of each phase."""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch_semantic_segmentation_amd as tssa
from torch_semantic_segmentation_amd import _native as N, ops
F = importlib.import_module('torch_semantic_segmentation_amd.models.fastscnn')
dev = 'cuda:0'
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
# (Cin, Cout, stride, H, W) of features.0.0, 0.1, 1.0, 1.1, 2.0, 2.1 at 2048 x 4096
for (cin, cout, s, H, W) in ((64, 64, 2, 256, 512), (64, 64, 1, 128, 256), (64, 96, 2, 128, 256), (96, 96, 1, 64, 128), (96, 128, 1, 64, 128), (128, 128, 1, 64, 128)):
    torch.manual_seed(0)
    m = F.BottleneckBlock(cin, cout, stride=s).to(dev).eval()
    tssa.set_compute_dtype(m, torch.bfloat16)
    x = ops.to_nhwc(torch.randn(1, cin, H, W, device=dev).to(torch.bfloat16))
    sh = ops.WeightShadows(m); sh.refresh()
    res = {}
    with torch.no_grad(), sh:
        for fused in (True, False):
            ops.eval_bottleneck = fused
            res[fused] = timeit(lambda: m(x))
        ops.eval_bottleneck = True
    line = 'bneck %3d->%3d->%3d s%d  %dx%d   one kernel %6.1f us   layer by layer %6.1f us' % (cin, 6 * cin, cout, s, H, W, res[True], res[False])
    if os.environ.get('TSS_TIMING') == '1':
        buf = (ctypes.c_ulonglong * 8)()
        N.lib().tss_debug_bk_timing(buf, 1)
        with torch.no_grad(), sh:
            m(x); torch.cuda.synchronize()
        N.lib().tss_debug_bk_timing(buf, 1)
        n = max(buf[7], 1)
        line += '   cycles/block: setup %d  stage %d  depthwise %d  project %d  expand %d  epilogue %d  (blocks %d)' % tuple([buf[q] // n for q in range(6)] + [n])
    print(line, flush=True)
