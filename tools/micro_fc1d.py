#!/usr/bin/env python
"""Microbenchmark (GPU box): the three-tap (1x3 / 3x1) layers of LEDNet at the benchmark's sizes -- forward, backward-data (with and
without a BatchNorm behind the layer) and the weight gradient (unfold + pointwise kernel), lean kernels (csrc/fc1d.hip) against the
generic implicit-GEMM kernels (TSS_OPT_DISABLE_FAST_PATHS), on rotating buffers; prints times, GB/s and the agreement of the two."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_semantic_segmentation_amd import _native as N, ops

dev = 'cuda:0'
S = N.stat_slabs()
BF = N.TSS_BF16


def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


rl = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30)).item()
# (B, H, W, C, axis, dil): layer1 / layer2 / layer3 of LEDNet's encoder at 8 x 3 x 1024 x 2048
CASES = ((8, 512, 1024, 16, 0, 1), (8, 512, 1024, 16, 1, 1), (8, 256, 512, 32, 0, 1), (8, 256, 512, 32, 1, 1),
         (8, 128, 256, 64, 0, 1), (8, 128, 256, 64, 1, 1), (8, 128, 256, 64, 0, 9), (8, 128, 256, 64, 1, 17))
if len(sys.argv) > 1 and sys.argv[1] == '--small':
    CASES = tuple((2, 64, 128) + c[3:] for c in CASES)
for (B, H, W, C, axis, dil) in CASES:
    P = B * H * W
    nset = max(1, min(6, int(800e6 // (P * C * 2 * 4))))
    mk = lambda: ops.new_nhwc(B, C, H, W, torch.bfloat16, dev).normal_()
    xs, ys, es, eins = [mk() for _ in range(nset)], [mk() for _ in range(nset)], [mk() for _ in range(nset)], [mk() for _ in range(nset)]
    v = lambda s=0.1: torch.randn(C, device=dev) * s
    mean, sc, bias = v(), torch.rand(C, device=dev) + 0.5, v()
    ga, gb, gce, gmu = torch.rand(C, device=dev) + 0.5, v(0.05), v(0.01), v()
    w = torch.randn(C, C, 3, device=dev) * 0.2
    w_tnc, w_tcn = torch.empty(3, C, C, device=dev), torch.empty(3, C, C, device=dev)
    st = N.stream()
    N.call('tss_permute_wtaps', N.ptr(w), N.ptr(w_tnc), N.ptr(w_tcn), C, C, 3, st)
    stats = torch.empty(S, 2 * C, dtype=torch.float64, device=dev)
    bst = torch.empty(S, 2 * C, dtype=torch.float64, device=dev)
    dw = torch.zeros(C, C, 3, device=dev)
    col = torch.empty(P, 3 * C, dtype=torch.bfloat16, device=dev)
    nws = N.lib().tss_pwconv_bwd_weight_ws(P, 3 * C, C, BF)
    ws = torch.empty(max(nws, 1), device=dev)
    xargs = lambda x: (N.ptr(x), C, N.ptr(mean), N.ptr(sc), N.ptr(bias), 1)
    gargs = lambda e, y: (N.ptr(e), C, N.ptr(y), C, N.ptr(ga), N.ptr(gb), N.ptr(gce), N.ptr(gmu))
    gplain = lambda e: (N.ptr(e), C, None, 0, None, None, None, None)

    def fwd():
        for x, y in zip(xs, ys):
            N.call('tss_conv1d3_fwd', *xargs(x), N.ptr(w_tnc), None, N.ptr(y), C, N.ptr(stats), B, H, W, C, C, axis, dil, BF, st)

    def bwd_y():
        for e, y, x, ei in zip(es, ys, xs, eins):
            N.call('tss_conv1d3_bwd_data', *gargs(e, y), N.ptr(w_tcn), *xargs(x), N.ptr(ei), C, N.ptr(bst), B, H, W, C, C, axis, dil, BF, st)

    def bwd_plain():
        for e, x, ei in zip(es, xs, eins):
            N.call('tss_conv1d3_bwd_data', *gplain(e), N.ptr(w_tcn), N.ptr(x), C, None, None, None, 1, N.ptr(ei), C, None, B, H, W, C, C,
                   axis, dil, BF, st)

    def wg_unfold():
        for e, y, x in zip(es, ys, xs):
            N.call('tss_im2col1d3', *xargs(x), N.ptr(col), B, H, W, C, axis, dil, BF, st)
            N.call('tss_pwconv_bwd_weight', *gargs(e, y), N.ptr(col), 3 * C, None, None, None, 0, N.ptr(dw), N.ptr(ws) if nws else None, 0,
                   P, 3 * C, C, BF, None, st)

    def wg_generic():
        for e, y, x in zip(es, ys, xs):
            N.call('tss_conv1d3_bwd_weight', *gargs(e, y), *xargs(x), N.ptr(dw), B, H, W, C, C, axis, dil, BF, st)

    res = {}
    for tag, fn, out, alg in (('fwd', fwd, lambda: (ys[0].clone(), stats.sum(0)), P * C * 2 * 2),
                              ('bwd(e,y)', bwd_y, lambda: (eins[0].clone(), bst.sum(0)), P * C * 2 * 4),
                              ('bwd(e)', bwd_plain, lambda: (eins[0].clone(),), P * C * 2 * 3)):
        keep = ys[0].clone() if tag == 'fwd' else None
        fn(); torch.cuda.synchronize()
        lean = out()
        t1 = timeit(fn) / nset
        N.call('tss_set_option', 1, 1)
        try:
            fn(); torch.cuda.synchronize()
            gen = out()
            t0 = timeit(fn, 3) / nset
        finally:
            N.call('tss_set_option', 1, 0)
        if keep is not None:
            ys[0].copy_(keep)
        print('%-9s %dx%dx%d C=%d axis=%d dil=%2d  lean %7.1f us (%5.0f GB/s)  generic %7.1f us   rel %s' % (
            tag, B, H, W, C, axis, dil, t1, alg / t1 / 1e3, t0, ' '.join('%.2e' % rl(a, b) for a, b in zip(lean, gen))), flush=True)
    rows = N.lib().tss_conv1d3_bwd_weight_rows(P, C, C, BF)
    wsr = torch.empty(max(rows, 1), 3 * C * C, device=dev)

    def wg_sweep():
        for e, y, x in zip(es, ys, xs):
            N.call('tss_conv1d3_bwd_weight_sweep', *gargs(e, y), *xargs(x), N.ptr(wsr), B, H, W, C, C, axis, dil, BF, st)
            ops._reduce_rows_now(wsr, dw, 3 * C * C, rows)

    def wg_sweep_plain():
        for e, x in zip(es, xs):
            N.call('tss_conv1d3_bwd_weight_sweep', *gplain(e), N.ptr(x), C, None, None, None, 1, N.ptr(wsr), B, H, W, C, C, axis, dil, BF, st)
            ops._reduce_rows_now(wsr, dw, 3 * C * C, rows)
    if rows:
        dw.zero_(); wg_sweep(); torch.cuda.synchronize(); d2 = dw.clone() / nset
        t2 = timeit(wg_sweep) / nset
        t3 = timeit(wg_sweep_plain) / nset
        dw.zero_(); wg_generic(); torch.cuda.synchronize(); d0 = dw.clone() / nset
        print('%-9s %dx%dx%d C=%d axis=%d dil=%2d  one sweep %7.1f us (%5.0f GB/s), no BatchNorm behind %7.1f us (%5.0f GB/s)  rows %d  rel %.2e' % (
            'wgrad', B, H, W, C, axis, dil, t2, P * C * 2 * 3 / t2 / 1e3, t3, P * C * 2 * 2 / t3 / 1e3, rows, rl(d2, d0)), flush=True)
    dw.zero_(); wg_unfold(); torch.cuda.synchronize(); d1 = dw.clone() / nset
    t1 = timeit(wg_unfold) / nset
    dw.zero_(); wg_generic(); torch.cuda.synchronize(); d0 = dw.clone() / nset
    t0 = timeit(wg_generic, 3) / nset
    print('%-9s %dx%dx%d C=%d axis=%d dil=%2d  unfold+pw %7.1f us (%5.0f GB/s)  generic %7.1f us   rel %.2e' % (
        'wgrad', B, H, W, C, axis, dil, t1, P * C * 2 * 3 / t1 / 1e3, t0, rl(d1, d0)), flush=True)
    del xs, ys, es, eins, col
    torch.cuda.empty_cache()
