#!/usr/bin/env python
"""Microbenchmark (GPU box): backward of the benchmark's LARGE 1x1 layers, one sweep (csrc/pwsweep.hip) against the pair
tss_pwconv_bwd_weight + tss_pwconv_bwd_data (+ _radd), on rotating buffers; prints times and the agreement of the two paths."""
import os, sys
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

# (B, H, W, Cin, Cout, x_pending, in_relu, radd): classifier / fusion pointwise, bottleneck expand (block input, with and without a
# skip gradient) and project at 1/8 and 1/16 resolution
CASES = ((8, 128, 256, 128, 128, 1, 0, 0), (8, 128, 256, 128, 128, 1, 1, 0), (8, 128, 256, 64, 384, 0, 0, 0),
         (8, 64, 128, 64, 384, 0, 0, 1), (8, 64, 128, 384, 64, 1, 1, 0), (4, 64, 128, 128, 128, 0, 0, 1))
if len(sys.argv) > 1 and sys.argv[1] == '--small':
    CASES = tuple((2, 128, 128) + c[3:] for c in CASES)
for (B, H, W, Cin, Cout, xp, relu, has_radd) in CASES:
    P = B * H * W
    nset = max(1, min(6, int(600e6 // (P * (2 * Cin + 2 * Cout) * 2))))
    mk = lambda c: ops.new_nhwc(B, c, H, W, torch.bfloat16, dev).normal_()
    es, ys, xs, eins = [mk(Cout) for _ in range(nset)], [mk(Cout) for _ in range(nset)], [mk(Cin) for _ in range(nset)], [mk(Cin) for _ in range(nset)]
    radd = mk(Cin) if has_radd else None
    v = lambda c, s=0.1: torch.randn(c, device=dev) * s
    ga, gb, gce, gmu = torch.rand(Cout, device=dev) + 0.5, v(Cout, 0.05), v(Cout, 0.01), v(Cout)
    mean, sc, bias = (v(Cin), torch.rand(Cin, device=dev) + 0.5, v(Cin)) if xp else (None, None, None)
    w = torch.randn(Cout, Cin, device=dev) * 0.2
    wT = w.t().contiguous().to(torch.bfloat16)
    dw = torch.zeros(Cout, Cin, device=dev)
    bst = torch.empty(S, 2 * Cin, dtype=torch.float64, device=dev) if xp else None
    assert N.lib().tss_pwconv_bwd_sweep_preferred(P, Cin, Cout, xp, N.TSS_BF16), (P, Cin, Cout, xp)
    rows = N.lib().tss_pwconv_bwd_sweep_rows(P, Cin, Cout)
    wsf = torch.empty(rows, Cout * Cin, device=dev)
    nws = N.lib().tss_pwconv_bwd_weight_ws(P, Cin, Cout, N.TSS_BF16)
    wsp = torch.empty(max(nws, 1), device=dev)
    st = N.stream()
    gargs = lambda e, y: (N.ptr(e), Cout, N.ptr(y), Cout, N.ptr(ga), N.ptr(gb), N.ptr(gce), N.ptr(gmu))
    xargs = lambda x: (N.ptr(x), Cin, N.ptr(mean), N.ptr(sc), N.ptr(bias), relu)
    def sweep():
        for e, y, x, ei in zip(es, ys, xs, eins):
            N.call('tss_pwconv_bwd_sweep', *gargs(e, y), N.ptr(wT), *xargs(x), xp, N.ptr(radd), Cin if has_radd else 0,
                   N.ptr(ei), Cin, N.ptr(bst), N.ptr(wsf), P, Cin, Cout, N.TSS_BF16, st)
            ops._reduce_rows_now(wsf, dw, Cout * Cin, rows)
    def pair():
        for e, y, x, ei in zip(es, ys, xs, eins):
            N.call('tss_pwconv_bwd_weight', *gargs(e, y), *xargs(x), N.ptr(dw), N.ptr(wsp) if nws else None, 1 if nws else 0, P, Cin, Cout,
                   N.TSS_BF16, None, st)
            red = (N.ptr(wsp) if nws else None, N.ptr(dw) if nws else None, 0, 0, 0)
            if has_radd:
                N.call('tss_pwconv_bwd_data_radd', *gargs(e, y), N.ptr(w), N.ptr(wT), N.ptr(ei), Cin, *red, N.ptr(radd), Cin, P, Cin, Cout, N.TSS_BF16, st)
            else:
                margs = xargs(x) if xp else (None, 0, None, None, None, 0)
                N.call('tss_pwconv_bwd_data', *gargs(e, y), N.ptr(w), N.ptr(wT), *margs, N.ptr(ei), Cin, N.ptr(bst), *red, P, Cin, Cout, N.TSS_BF16, st)
    rl = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    dw.zero_(); pair(); torch.cuda.synchronize()
    ei0, dw0, st0 = eins[0].float().clone(), dw.clone(), (bst.sum(0) if xp else None)
    t0 = timeit(pair) / nset
    dw.zero_(); sweep(); torch.cuda.synchronize()
    ei1, dw1, st1 = eins[0].float().clone(), dw.clone(), (bst.sum(0) if xp else None)
    t1 = timeit(sweep) / nset
    alg = P * (2 * Cout + (3 if has_radd else 2) * Cin) * 2
    print('pw bwd  %dx%dx%d  %3d->%3d xp=%d relu=%d radd=%d  one sweep %7.1f us (%5.0f GB/s)  pair %7.1f us   e_in rel %.2e (max abs %.2e)  dW rel %.2e  stats rel %s' % (
        B, H, W, Cin, Cout, xp, relu, has_radd, t1, alg / t1 / 1e3, t0, rl(ei1, ei0), (ei1 - ei0).abs().max().item(), rl(dw1, dw0),
        ('%.2e' % rl(st1, st0)) if xp else '-'), flush=True)
    del es, ys, xs, eins
    torch.cuda.empty_cache()
