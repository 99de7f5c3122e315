#!/usr/bin/env python
"""Microbenchmark (GPU box): the classifier convolution (128 -> 19 classes, 8 x 128 x 256 pixels) with nn.Dropout applied on load
(tss_pwconv_fwd_drop / tss_pwconv_bwd_fused_drop) against the same kernels without dropout, stand-alone on rotating buffers."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_semantic_segmentation_amd import _native as N, ops  # noqa: E402

dev = 'cuda:0'


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


torch.manual_seed(0)
B, H, W, K, NC = 8, 128, 256, 128, 19
P = B * H * W
nset = 6
st = N.stream()
xs = [ops.new_nhwc(B, K, H, W, torch.bfloat16, dev).normal_() for _ in range(nset)]
ys = [ops.new_nhwc(B, NC, H, W, torch.bfloat16, dev) for _ in range(nset)]
es = [ops.new_nhwc(B, NC, H, W, torch.bfloat16, dev).normal_() for _ in range(nset)]
eins = [ops.new_nhwc(B, K, H, W, torch.bfloat16, dev) for _ in range(nset)]
w = torch.randn(NC, K, device=dev) * 0.1
bias = torch.randn(NC, device=dev) * 0.1
mean, sc, bb = torch.randn(K, device=dev) * 0.1, torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
mask = torch.empty((P, 16), dtype=torch.uint8, device=dev)
counter = torch.tensor([1234], dtype=torch.int64, device=dev)
bst = torch.empty(N.stat_slabs(), 2 * K, dtype=torch.float64, device=dev)
rows = max(N.lib().tss_pwconv_bwd_fused_rows(P, K, NC), N.lib().tss_pwconv_bwd_fused_drop_rows(P))
ws = torch.empty(rows, NC * K, dtype=torch.float32, device=dev)
bws = torch.empty(rows, NC, dtype=torch.float32, device=dev)
N.call('tss_dropout_mask', N.ptr(counter), N.ptr(mask), P, K, 0.1, st)


def fwd(drop):
    for x, y in zip(xs, ys):
        if drop:
            N.call('tss_pwconv_fwd_drop', N.ptr(x), ops.ld(x), N.ptr(mean), N.ptr(sc), N.ptr(bb), 1, N.ptr(w), None, N.ptr(bias),
                   N.ptr(y), ops.ld(y), N.ptr(mask), 0.1, N.ptr(counter), P, K, NC, N.TSS_BF16, st)
        else:
            N.call('tss_pwconv_fwd', N.ptr(x), ops.ld(x), N.ptr(mean), N.ptr(sc), N.ptr(bb), 1, N.ptr(w), None, N.ptr(bias),
                   N.ptr(y), ops.ld(y), None, P, K, NC, N.TSS_BF16, st)


def bwd(drop):
    for x, e, ei in zip(xs, es, eins):
        if drop:
            N.call('tss_pwconv_bwd_fused_drop', N.ptr(e), ops.ld(e), None, 0, None, None, None, None, N.ptr(w), N.ptr(x), ops.ld(x),
                   N.ptr(mean), N.ptr(sc), N.ptr(bb), 1, 1, N.ptr(mask), 0.1, N.ptr(ei), ops.ld(ei), N.ptr(bst), N.ptr(ws), N.ptr(bws),
                   P, K, NC, N.TSS_BF16, st)
        else:
            N.call('tss_pwconv_bwd_fused', N.ptr(e), ops.ld(e), None, 0, None, None, None, None, N.ptr(w), None, N.ptr(x), ops.ld(x),
                   N.ptr(mean), N.ptr(sc), N.ptr(bb), 1, 1, N.ptr(ei), ops.ld(ei), N.ptr(bst), N.ptr(ws), N.ptr(bws),
                   P, K, NC, N.TSS_BF16, st)


print('mask kernel            %7.1f us' % timeit(lambda: N.call('tss_dropout_mask', N.ptr(counter), N.ptr(mask), P, K, 0.1, st)))
print('forward   plain %7.1f us   dropout on load %7.1f us' % (timeit(lambda: fwd(False)) / nset, timeit(lambda: fwd(True)) / nset))
print('backward  plain %7.1f us   dropout on load %7.1f us' % (timeit(lambda: bwd(False)) / nset, timeit(lambda: bwd(True)) / nset))
