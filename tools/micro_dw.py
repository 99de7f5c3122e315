#!/usr/bin/env python
"""Microbenchmark (GPU box): the depthwise 3x3 entry points on the benchmark's own layer shapes, stand-alone, on a rotating
set of buffers (cold caches, as inside a step).  Prints time, algorithmic GB/s and a checksum of every result so two builds /
two settings of an A/B switch (TSS_DW_ROLL=0|1) can be compared line by line: the outputs must agree bit for bit."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_semantic_segmentation_amd import _native as N, ops  # noqa: E402

dev = 'cuda:0'
which = sys.argv[1] if len(sys.argv) > 1 else 'fwd'


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


def digest(t):
    return hashlib.sha1(t.detach().contiguous().cpu().view(torch.uint8).numpy().tobytes()).hexdigest()[:12]


S = N.stat_slabs()
# (B, C, Hin, Win, stride): FastSCNN's depthwise layers at 8 x 3 x 1024 x 2048, then odd shapes
LAYERS = ((8, 32, 512, 1024, 2), (8, 48, 256, 512, 2), (8, 384, 128, 256, 2), (8, 384, 64, 128, 1), (8, 384, 64, 128, 2),
          (8, 576, 32, 64, 1), (8, 768, 32, 64, 1), (8, 128, 128, 256, 1), (2, 24, 37, 53, 1), (2, 200, 37, 53, 2), (1, 8, 5, 7, 1))
torch.manual_seed(0)
for (B, C, H, W, s) in LAYERS:
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    nset = max(1, min(8, int(600e6 // (B * C * H * W * 2))))
    xs = [ops.new_nhwc(B, C, H, W, torch.bfloat16, dev).normal_() for _ in range(nset)]
    ys = [ops.new_nhwc(B, C, Ho, Wo, torch.bfloat16, dev) for _ in range(nset)]
    w = torch.randn(C, 9, device=dev) * 0.3
    mean = torch.randn(C, device=dev) * 0.1
    sc = torch.rand(C, device=dev) + 0.5
    bias = torch.randn(C, device=dev) * 0.1
    stats = torch.empty(S, 2 * C, dtype=torch.float64, device=dev)
    st = N.stream()
    if which == 'fwd':
        def run():
            for x, y in zip(xs, ys):
                N.call('tss_dwconv3x3_fwd', N.ptr(x), ops.ld(x), N.ptr(mean), N.ptr(sc), N.ptr(bias), 1, N.ptr(w), N.ptr(y), ops.ld(y),
                       N.ptr(stats), B, H, W, C, s, 1, N.TSS_BF16, st)
        os.environ['TSS_DW_ROLL'] = '0'
        run(); torch.cuda.synchronize()
        y_ref, st_ref = ys[0].clone(), stats.sum(0)
        t_ref = timeit(run) / nset
        os.environ['TSS_DW_ROLL'] = '1'
        run(); torch.cuda.synchronize()
        d = (ys[0].float() - y_ref.float()).abs()
        nz = ((ys[0].view(torch.int16) != y_ref.view(torch.int16)) & (d == 0)).sum().item()
        print('   vs strip kernel: %d of %d values differ (max abs %.3e), %d differ only in the sign of zero; stats rel %.2e; strip %.1f us' % (
            (d > 0).sum().item(), d.numel(), d.max().item(), nz, ((stats.sum(0) - st_ref).abs().max() / st_ref.abs().max()).item(), t_ref))
        if B * C * H * W <= 30e6:     # both against an f64 evaluation of the same layer (rounded to bf16 once)
            a64 = torch.relu(xs[0].double() * sc.double()[None, :, None, None] + torch.addcmul(bias, mean, sc, value=-1).double()[None, :, None, None])
            y64 = torch.nn.functional.conv2d(a64, w.double().view(C, 1, 3, 3), stride=s, padding=1, groups=C)
            r = y64.to(torch.bfloat16)
            print('   against f64: strip differs in %d values, row-pipelined in %d' % (
                (y_ref.float() != r.float()).sum().item(), (ys[0].float() != r.float()).sum().item()))
            del a64, y64, r
        t = timeit(run) / nset
        alg = (B * C * H * W + B * C * Ho * Wo) * 2
        print('dw fwd  %dx%dx%dx%d s%d  %7.1f us  %6.0f GB/s   y %s  stats %s  sum %.6e' % (
            B, C, H, W, s, t, alg / t / 1e3, digest(ys[0]), digest(stats.sum(0).float()), stats.sum(0)[:C].sum().item()))
    if which == 'bwd':
        es = [ops.new_nhwc(B, C, Ho, Wo, torch.bfloat16, dev).normal_() for _ in range(nset)]
        yr = [ops.new_nhwc(B, C, Ho, Wo, torch.bfloat16, dev).normal_() for _ in range(nset)]
        eins = [ops.new_nhwc(B, C, H, W, torch.bfloat16, dev) for _ in range(nset)]
        ga = torch.rand(C, device=dev) + 0.5; gb = torch.randn(C, device=dev) * 0.05; gce = torch.randn(C, device=dev) * 0.01
        gmu = torch.randn(C, device=dev) * 0.1
        ws = torch.empty(S, C * 9, device=dev)
        dwt = torch.zeros(C, 9, device=dev)
        bst = torch.empty(S, 2 * C, dtype=torch.float64, device=dev)

        def fused():
            for x, e, y, ei in zip(xs, es, yr, eins):
                N.call('tss_dwconv3x3_bwd_fused', N.ptr(e), ops.ld(e), N.ptr(y), ops.ld(y), N.ptr(ga), N.ptr(gb), N.ptr(gce), N.ptr(gmu),
                       N.ptr(w), N.ptr(x), ops.ld(x), N.ptr(mean), N.ptr(sc), N.ptr(bias), 1, 1, N.ptr(ei), ops.ld(ei), N.ptr(bst),
                       N.ptr(ws), N.ptr(dwt), B, H, W, C, s, 1, N.TSS_BF16, st)

        def pair():
            for x, e, y, ei in zip(xs, es, yr, eins):
                N.call('tss_dwconv3x3_bwd_weight', N.ptr(e), ops.ld(e), N.ptr(y), ops.ld(y), N.ptr(ga), N.ptr(gb), N.ptr(gce), N.ptr(gmu),
                       N.ptr(x), ops.ld(x), N.ptr(mean), N.ptr(sc), N.ptr(bias), 1, N.ptr(dwt), N.ptr(ws), 1, B, H, W, C, s, 1, N.TSS_BF16, st)
                N.call('tss_dwconv3x3_bwd_data', N.ptr(e), ops.ld(e), N.ptr(y), ops.ld(y), N.ptr(ga), N.ptr(gb), N.ptr(gce), N.ptr(gmu),
                       N.ptr(w), N.ptr(x), ops.ld(x), N.ptr(mean), N.ptr(sc), N.ptr(bias), 1, N.ptr(ei), ops.ld(ei), N.ptr(bst),
                       N.ptr(ws), N.ptr(dwt), B, H, W, C, s, 1, N.TSS_BF16, st)
        dwt.zero_(); pair(); torch.cuda.synchronize()
        ei0, dw0, st0 = eins[0].float().clone(), dwt.clone() / nset, bst.sum(0)
        t0 = timeit(pair) / nset
        dwt.zero_(); fused(); torch.cuda.synchronize()
        ei1, dw1, st1 = eins[0].float().clone(), dwt.clone() / nset, bst.sum(0)
        t1 = timeit(fused) / nset
        rl = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
        alg = (2 * B * C * H * W + 2 * B * C * Ho * Wo) * 2
        print('dw bwd  %dx%dx%dx%d s%d  one sweep %7.1f us (%5.0f GB/s on e, y, x, e_in once)  pair %7.1f us   e_in rel %.2e  dW rel %.2e  stats rel %.2e' % (
            B, C, H, W, s, t1, alg / t1 / 1e3, t0, rl(ei1, ei0), rl(dw1, dw0), rl(st1, st0)))
        del es, yr, eins
    del xs, ys
    torch.cuda.empty_cache()
