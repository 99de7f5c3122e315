#!/usr/bin/env python
"""Microbenchmark (GPU box): the streaming kernels (join forward / backward, copy) on tensors larger than the 256 MiB Infinity
Cache, stand-alone -- what the access pattern itself sustains, against what the same kernels reach inside a training step."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_semantic_segmentation_amd import _native as N, ops

dev = 'cuda:0'
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

S = N.stat_slabs()
for (B, C, H, W) in ((8, 128, 128, 256), (8, 128, 256, 512), (8, 32, 512, 1024), (8, 384, 128, 256)):
    mk = lambda: ops.new_nhwc(B, C, H, W, torch.bfloat16, dev).normal_()
    a, b, out, dout, e = mk(), mk(), mk(), mk(), mk()
    mean = torch.zeros(C, device=dev); sc = torch.ones(C, device=dev)
    sa = torch.empty(S, 2 * C, dtype=torch.float64, device=dev); sb = torch.empty_like(sa)
    P, st = B * H * W, N.stream()
    mb = a.numel() * 2 / 1e6
    t = timeit(lambda: N.call('tss_join_fwd', N.ptr(a), C, N.ptr(mean), N.ptr(sc), N.ptr(mean), N.ptr(b), C, N.ptr(mean), N.ptr(sc), N.ptr(mean), N.ptr(out), C, 1, 0.0, None, P, C, 1, st))
    print('join_fwd 2 in  %dx%dx%dx%d  %6.0f MB/tensor  %7.1f us  %6.0f GB/s' % (B, C, H, W, mb, t, 3 * mb / t * 1e3))
    t = timeit(lambda: N.call('tss_join_fwd', N.ptr(a), C, N.ptr(mean), N.ptr(sc), N.ptr(mean), None, 0, None, None, None, N.ptr(out), C, 1, 0.0, None, P, C, 1, st))
    print('join_fwd 1 in  %dx%dx%dx%d  %6.0f MB/tensor  %7.1f us  %6.0f GB/s' % (B, C, H, W, mb, t, 2 * mb / t * 1e3))
    t = timeit(lambda: N.call('tss_join_bwd', N.ptr(dout), C, N.ptr(out), C, 1, N.ptr(a), C, N.ptr(mean), N.ptr(sa), N.ptr(b), C, N.ptr(mean), N.ptr(sb), N.ptr(e), C, 1.0, P, C, 1, st))
    print('join_bwd 4r 1w %dx%dx%dx%d  %6.0f MB/tensor  %7.1f us  %6.0f GB/s' % (B, C, H, W, mb, t, 5 * mb / t * 1e3))
    t = timeit(lambda: out.copy_(a))
    print('torch copy     %dx%dx%dx%d  %6.0f MB/tensor  %7.1f us  %6.0f GB/s' % (B, C, H, W, mb, t, 2 * mb / t * 1e3))
    t = timeit(lambda: N.call('tss_copy_nhwc', N.ptr(a), C, N.ptr(out), C, P, C, 1, st))
    print('tss_copy_nhwc  %dx%dx%dx%d  %6.0f MB/tensor  %7.1f us  %6.0f GB/s' % (B, C, H, W, mb, t, 2 * mb / t * 1e3))
    del a, b, out, dout, e
    torch.cuda.empty_cache()

# the same join on a ROTATING working set (many distinct buffers, as inside a training step where every kernel touches tensors
# nobody has touched for milliseconds): separates "cold TLB / cold cache" from the kernel itself
B, C, H, W = 8, 128, 128, 256
for nsets in (1, 4, 16, 48):
    sets = [[ops.new_nhwc(B, C, H, W, torch.bfloat16, dev).normal_() for _ in range(3)] for _ in range(nsets)]
    mean = torch.zeros(C, device=dev); sc = torch.ones(C, device=dev)
    P, st = B * H * W, N.stream()
    mb = sets[0][0].numel() * 2 / 1e6
    def sweep():
        for a, b, out in sets:
            N.call('tss_join_fwd', N.ptr(a), C, N.ptr(mean), N.ptr(sc), N.ptr(mean), N.ptr(b), C, N.ptr(mean), N.ptr(sc), N.ptr(mean), N.ptr(out), C, 1, 0.0, None, P, C, 1, st)
    t = timeit(sweep, n=5) / nsets
    print('join_fwd 2 in, %2d rotating sets (%5.0f MB working set): %7.1f us per launch  %6.0f GB/s' % (nsets, 3 * mb * nsets, t, 3 * mb / t * 1e3))
    del sets
    torch.cuda.empty_cache()
