#!/usr/bin/env python
"""Microbenchmark (GPU box): bilinear upsample + dilation-4 depthwise 3x3 of the feature-fusion modules at the benchmark's size
(8 x 128 x 32 x 64 -> 128 x 256), as ONE operator (csrc/updw.hip) and as the two operators it replaces (bilinear_nhwc + the
strip kernels of dwconv.hip), forward and backward, stand-alone on a rotating set of buffers."""
import ctypes
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


S = N.stat_slabs()
torch.manual_seed(0)
for (B, C, Hs, Ws, Ho, Wo, D) in ((8, 128, 32, 64, 128, 256, 4), (4, 128, 16, 32, 64, 128, 4), (1, 128, 64, 128, 256, 512, 4)):
    nset = 6
    st = N.stream()
    xs = [ops.new_nhwc(B, C, Hs, Ws, torch.bfloat16, dev).normal_() for _ in range(nset)]
    ups = [ops.new_nhwc(B, C, Ho, Wo, torch.bfloat16, dev) for _ in range(nset)]
    ys = [ops.new_nhwc(B, C, Ho, Wo, torch.bfloat16, dev) for _ in range(nset)]
    es = [ops.new_nhwc(B, C, Ho, Wo, torch.bfloat16, dev).normal_() for _ in range(nset)]
    eups = [ops.new_nhwc(B, C, Ho, Wo, torch.bfloat16, dev) for _ in range(nset)]
    dxs = [ops.new_nhwc(B, C, Hs, Ws, torch.bfloat16, dev) for _ in range(nset)]
    w = torch.randn(C, 9, device=dev) * 0.3
    ga, gb, gce, gmu = (torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1,
                        torch.randn(C, device=dev) * 0.1)
    stats = torch.empty(S, 2 * C, dtype=torch.float64, device=dev)
    ws = torch.empty(max(S, N.lib().tss_updw_ws_rows(B, Hs, Ws, Ho, Wo, C, D, N.TSS_BF16)), C * 9, dtype=torch.float32, device=dev)
    dw = torch.zeros(C, 9, device=dev)
    tmp = torch.empty(B * Hs * Wo * C, dtype=torch.float32, device=dev)
    rows = ctypes.c_int(0)

    def fused_fwd():
        for x, y in zip(xs, ys):
            N.call('tss_updw_fwd', N.ptr(x), ops.ld(x), Hs, Ws, N.ptr(w), N.ptr(y), ops.ld(y), N.ptr(stats), B, Ho, Wo, C, D, N.TSS_BF16, st)

    def split_fwd():
        for x, u, y in zip(xs, ups, ys):
            N.call('tss_bilinear_nhwc_fwd', N.ptr(x), ops.ld(x), N.ptr(u), ops.ld(u), B, Hs, Ws, Ho, Wo, C, N.TSS_BF16, st)
            N.call('tss_dwconv3x3_fwd', N.ptr(u), ops.ld(u), None, None, None, 0, N.ptr(w), N.ptr(y), ops.ld(y), N.ptr(stats),
                   B, Ho, Wo, C, 1, D, N.TSS_BF16, st)

    def fused_bwd():
        for x, y, e, eu, dx in zip(xs, ys, es, eups, dxs):
            N.call('tss_updw_bwd', N.ptr(e), ops.ld(e), N.ptr(y), ops.ld(y), N.ptr(ga), N.ptr(gb), N.ptr(gce), N.ptr(gmu), N.ptr(w),
                   N.ptr(x), ops.ld(x), Hs, Ws, N.ptr(eu), ops.ld(eu), N.ptr(ws), B, Ho, Wo, C, D, N.TSS_BF16, st, ctypes.byref(rows))
            N.call('tss_bilinear_nhwc_bwd', N.ptr(eu), ops.ld(eu), N.ptr(dx), ops.ld(dx), N.ptr(tmp), B, Hs, Ws, Ho, Wo, C, N.TSS_BF16, st)

    def split_bwd():
        for u, y, e, eu, dx in zip(ups, ys, es, eups, dxs):
            N.call('tss_dwconv3x3_bwd_weight', N.ptr(e), ops.ld(e), N.ptr(y), ops.ld(y), N.ptr(ga), N.ptr(gb), N.ptr(gce), N.ptr(gmu),
                   N.ptr(u), ops.ld(u), None, None, None, 0, N.ptr(dw), N.ptr(ws), 1, B, Ho, Wo, C, 1, D, N.TSS_BF16, st)
            N.call('tss_dwconv3x3_bwd_data', N.ptr(e), ops.ld(e), N.ptr(y), ops.ld(y), N.ptr(ga), N.ptr(gb), N.ptr(gce), N.ptr(gmu),
                   N.ptr(w), None, 0, None, None, None, 0, N.ptr(eu), ops.ld(eu), None, N.ptr(ws), N.ptr(dw), B, Ho, Wo, C, 1, D, N.TSS_BF16, st)
            N.call('tss_bilinear_nhwc_bwd', N.ptr(eu), ops.ld(eu), N.ptr(dx), ops.ld(dx), N.ptr(tmp), B, Hs, Ws, Ho, Wo, C, N.TSS_BF16, st)

    split_fwd(); torch.cuda.synchronize()
    y_ref, s_ref = ys[0].clone(), stats.sum(0)
    fused_fwd(); torch.cuda.synchronize()
    dy = (ys[0].float() - y_ref.float()).abs()
    print('%dx%dx%dx%d -> %dx%d d%d: forward differs in %d of %d values (max abs %.3e), stats rel %.2e' % (
        B, C, Hs, Ws, Ho, Wo, D, (dy > 0).sum().item(), dy.numel(), dy.max().item(),
        ((stats.sum(0) - s_ref).abs().max() / s_ref.abs().max()).item()))
    split_bwd(); torch.cuda.synchronize()
    dx_ref = dxs[0].clone()
    fused_bwd(); torch.cuda.synchronize()
    ddx = (dxs[0].float() - dx_ref.float()).abs()
    print('   backward dx: rel %.2e (max abs %.3e of %.3e)' % (
        ((dxs[0].float() - dx_ref.float()).norm() / dx_ref.float().norm()).item(), ddx.max().item(), dx_ref.float().abs().max().item()))
    tf, ts = timeit(fused_fwd) / nset, timeit(split_fwd) / nset
    tfb, tsb = timeit(fused_bwd) / nset, timeit(split_bwd) / nset
    print('   forward  one operator %7.1f us   two operators %7.1f us' % (tf, ts))
    print('   backward one operator %7.1f us   two operators %7.1f us   (both + the transposed interpolation)' % (tfb, tsb))
