#!/usr/bin/env python
"""Microbenchmark (GPU box): pointwise conv kernels over a size sweep."""
import os, sys
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
for (K, Nn) in ((128, 128), (64, 384), (384, 64), (32, 48), (128, 768), (768, 128)):
    for P in (16384, 65536, 262144, 1048576):
        if P * (K + Nn) > 6e8: continue
        x = torch.randn(P, K, device=dev).bfloat16(); y = torch.empty(P, Nn, device=dev, dtype=torch.bfloat16)
        e = torch.randn(P, Nn, device=dev).bfloat16(); ein = torch.empty(P, K, device=dev, dtype=torch.bfloat16)
        w = torch.randn(Nn, K, device=dev) * 0.1
        dw = torch.zeros(Nn, K, device=dev)
        stats = torch.empty(S, 2 * Nn, dtype=torch.float64, device=dev); bst = torch.empty(S, 2 * K, dtype=torch.float64, device=dev)
        mK = torch.zeros(K, device=dev); sK = torch.ones(K, device=dev)
        mN = torch.zeros(Nn, device=dev); sN = torch.ones(Nn, device=dev)
        st = N.stream()
        fwd = lambda: N.call('tss_pwconv_fwd', N.ptr(x), K, N.ptr(mK), N.ptr(sK), N.ptr(mK), 1, N.ptr(w), None, None, N.ptr(y), Nn, N.ptr(stats), P, K, Nn, 1, st)
        fwd0 = lambda: N.call('tss_pwconv_fwd', N.ptr(x), K, None, None, None, 0, N.ptr(w), None, None, N.ptr(y), Nn, None, P, K, Nn, 1, st)
        bwd = lambda: N.call('tss_pwconv_bwd_data', N.ptr(e), Nn, N.ptr(y), Nn, N.ptr(sN), N.ptr(sN), N.ptr(mN), N.ptr(mN), N.ptr(w), None,
                             N.ptr(x), K, N.ptr(mK), N.ptr(sK), N.ptr(mK), 1, N.ptr(ein), K, N.ptr(bst), None, None, P, K, Nn, 1, st)
        wg = lambda: N.call('tss_pwconv_bwd_weight', N.ptr(e), Nn, N.ptr(y), Nn, N.ptr(sN), N.ptr(sN), N.ptr(mN), N.ptr(mN),
                            N.ptr(x), K, N.ptr(mK), N.ptr(sK), N.ptr(mK), 1, N.ptr(dw), None, 0, P, K, Nn, 1, st)
        tf, tf0, tb, tw = timeit(fwd), timeit(fwd0), timeit(bwd), timeit(wg)
        mbf = P * (K + Nn) * 2 / 1e6; mbb = P * (2 * Nn + 2 * K) * 2 / 1e6; mbw = P * (2 * Nn + K) * 2 / 1e6
        print('pw K=%3d N=%3d P=%8d | fwd %7.1f us %5.0f GB/s (plain %7.1f us %5.0f) | bwd_data %7.1f us %5.0f GB/s | wgrad %7.1f us %5.0f GB/s'
              % (K, Nn, P, tf, mbf / tf * 1e3, tf0, mbf / tf0 * 1e3, tb, mbb / tb * 1e3, tw, mbw / tw * 1e3))
