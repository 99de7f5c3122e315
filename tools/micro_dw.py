#!/usr/bin/env python
"""Microbenchmark (GPU box): time individual C-ABI kernels over a size sweep to separate fixed cost from bandwidth."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_semantic_segmentation_amd import _native as N, ops

dev = 'cuda:0'
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

S = N.stat_slabs()
for C in (32, 128, 384, 576):
    for (B, H, W) in ((8, 32, 64), (8, 64, 128), (8, 128, 256), (8, 256, 512)):
        if C * B * H * W > 4e8: continue
        for stride in (1, 2):
            x = ops.new_nhwc(B, C, H, W, torch.bfloat16, dev); x.normal_()
            Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
            y = ops.new_nhwc(B, C, Ho, Wo, torch.bfloat16, dev)
            w = torch.randn(C, 1, 3, 3, device=dev)
            stats = torch.empty(S, 2 * C, dtype=torch.float64, device=dev)
            mean = torch.zeros(C, device=dev); sc = torch.ones(C, device=dev)
            st = N.stream()
            f = lambda: N.call('tss_dwconv3x3_fwd', N.ptr(x), C, N.ptr(mean), N.ptr(sc), N.ptr(mean), 1, N.ptr(w), N.ptr(y), C, N.ptr(stats), B, H, W, C, stride, 1, 1, st)
            f_nostats = lambda: N.call('tss_dwconv3x3_fwd', N.ptr(x), C, None, None, None, 0, N.ptr(w), N.ptr(y), C, None, B, H, W, C, stride, 1, 1, st)
            t1, t2 = timeit(f), timeit(f_nostats)
            mb = (x.numel() + y.numel()) * 2 / 1e6
            print('dw_fwd C=%3d %dx%3dx%3d s%d  %7.1f MB  %7.1f us (%6.0f GB/s)   no-stats/no-affine %7.1f us (%6.0f GB/s)' % (C, B, H, W, stride, mb, t1, mb / t1 * 1e3, t2, mb / t2 * 1e3))
# copy kernel as the achievable-bandwidth yardstick
for n in (2**22, 2**24, 2**26, 2**28):
    a = torch.empty(n, dtype=torch.bfloat16, device=dev); b = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a))
    print('torch copy %6.1f MB  %7.1f us  %6.0f GB/s' % (2 * n * 2 / 1e6, t, 2 * n * 2 / 1e6 / t * 1e3))
    xa = a.view(1, -1, 1, 128).permute(0, 3, 1, 2)
    t = timeit(lambda: N.call('tss_copy_nhwc', N.ptr(a), 128, N.ptr(b), 128, n // 128, 128, 1, N.stream()))
    print('tss_copy   %6.1f MB  %7.1f us  %6.0f GB/s' % (2 * n * 2 / 1e6, t, 2 * n * 2 / 1e6 / t * 1e3))
