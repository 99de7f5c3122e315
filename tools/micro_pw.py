#!/usr/bin/env python
"""Microbenchmark (GPU box): backward of the benchmark's few-channel 1x1 layers, one sweep (csrc/pwbwd.hip) against the pair
tss_pwconv_bwd_weight + tss_pwconv_bwd_data, on rotating buffers; prints times and the agreement of the two paths."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_semantic_segmentation_amd import _native as N, ops

dev = 'cuda:0'
S = N.stat_slabs()
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

# (B, H, W, Cin, Cout): dsconv1 / dsconv2 pointwise, classifier dsconv pointwise, fusion, a bottleneck projection
for (B, H, W, Cin, Cout) in ((8, 256, 512, 32, 48), (8, 128, 256, 48, 64), (8, 128, 256, 128, 128), (8, 128, 256, 64, 128), (8, 64, 128, 64, 64)):
    P = B * H * W
    nset = max(1, min(6, int(500e6 // (P * (Cin + 2 * Cout) * 2))))
    mk = lambda c: ops.new_nhwc(B, c, H, W, torch.bfloat16, dev).normal_()
    es, ys, xs, eins = [mk(Cout) for _ in range(nset)], [mk(Cout) for _ in range(nset)], [mk(Cin) for _ in range(nset)], [mk(Cin) for _ in range(nset)]
    v = lambda c, s=0.1: torch.randn(c, device=dev) * s
    ga, gb, gce, gmu = torch.rand(Cout, device=dev) + 0.5, v(Cout, 0.05), v(Cout, 0.01), v(Cout)
    mean, sc, bias = v(Cin), torch.rand(Cin, device=dev) + 0.5, v(Cin)
    w = torch.randn(Cout, Cin, device=dev) * 0.2
    wT = w.t().contiguous().to(torch.bfloat16)
    dw = torch.zeros(Cout, Cin, device=dev)
    bst = torch.empty(S, 2 * Cin, dtype=torch.float64, device=dev)
    rows = N.lib().tss_pwconv_bwd_fused_rows(P, Cin, Cout)
    wsf = torch.empty(rows, Cout * Cin, device=dev)
    nws = N.lib().tss_pwconv_bwd_weight_ws(P, Cin, Cout, N.TSS_BF16)
    wsp = torch.empty(max(nws, 1), device=dev)
    st = N.stream()
    def fused():
        for e, y, x, ei in zip(es, ys, xs, eins):
            N.call('tss_pwconv_bwd_fused', N.ptr(e), Cout, N.ptr(y), Cout, N.ptr(ga), N.ptr(gb), N.ptr(gce), N.ptr(gmu), N.ptr(w), N.ptr(wT),
                   N.ptr(x), Cin, N.ptr(mean), N.ptr(sc), N.ptr(bias), 1, 1, N.ptr(ei), Cin, N.ptr(bst), N.ptr(wsf), None, P, Cin, Cout, N.TSS_BF16, st)
            ops._reduce_rows_now(wsf, dw, Cout * Cin, rows)
    def pair():
        for e, y, x, ei in zip(es, ys, xs, eins):
            N.call('tss_pwconv_bwd_weight', N.ptr(e), Cout, N.ptr(y), Cout, N.ptr(ga), N.ptr(gb), N.ptr(gce), N.ptr(gmu),
                   N.ptr(x), Cin, N.ptr(mean), N.ptr(sc), N.ptr(bias), 1, N.ptr(dw), N.ptr(wsp) if nws else None, 1 if nws else 0, P, Cin, Cout, N.TSS_BF16, None, st)
            N.call('tss_pwconv_bwd_data', N.ptr(e), Cout, N.ptr(y), Cout, N.ptr(ga), N.ptr(gb), N.ptr(gce), N.ptr(gmu), N.ptr(w), N.ptr(wT),
                   N.ptr(x), Cin, N.ptr(mean), N.ptr(sc), N.ptr(bias), 1, N.ptr(ei), Cin, N.ptr(bst),
                   N.ptr(wsp) if nws else None, N.ptr(dw) if nws else None, 0, 0, 0, P, Cin, Cout, N.TSS_BF16, st)
    rl = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    dw.zero_(); pair(); torch.cuda.synchronize()
    ei0, dw0, st0 = eins[0].float().clone(), dw.clone(), bst.sum(0)
    t0 = timeit(pair) / nset
    dw.zero_(); fused(); torch.cuda.synchronize()
    ei1, dw1, st1 = eins[0].float().clone(), dw.clone(), bst.sum(0)
    t1 = timeit(fused) / nset
    alg = P * (2 * Cout + 2 * Cin) * 2
    print('pw bwd  %dx%dx%d  %d->%d  one sweep %7.1f us (%5.0f GB/s on e, y, x, e_in once)  pair %7.1f us   e_in rel %.2e  dW rel %.2e  stats rel %.2e' % (
        B, H, W, Cin, Cout, t1, alg / t1 / 1e3, t0, rl(ei1, ei0), rl(dw1, dw0), rl(st1, st0)))
    del es, ys, xs, eins
    torch.cuda.empty_cache()
