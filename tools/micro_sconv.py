#!/usr/bin/env python
"""Microbenchmark (GPU box): the stride-2 dense 3x3 of the downsampling blocks (LEDNet / ESNet) -- forward, backward-data, weight gradient,
lean kernels (csrc/sconv.hip) against the generic implicit-GEMM kernels (TSS_OPT_DISABLE_FAST_PATHS); times, GB/s, agreement."""
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
# (B, Hin, Win, C): DownsamplingBlock 2 / 3 of LEDNet at 8 x 3 x 1024 x 2048; small ragged maps
CASES = ((8, 512, 1024, 32), (8, 256, 512, 64), (2, 37, 50, 32), (3, 22, 35, 64))
if len(sys.argv) > 1 and sys.argv[1] == '--small':
    CASES = CASES[2:]
for (B, H, W, C) in CASES:
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    P, Po = B * H * W, B * Ho * Wo
    x = ops.new_nhwc(B, C, H, W, torch.bfloat16, dev).normal_()
    z = ops.new_nhwc(B, 2 * C, Ho, Wo, torch.bfloat16, dev).normal_()       # the concat buffer: the convolution owns channels [0, C)
    e = ops.new_nhwc(B, 2 * C, Ho, Wo, torch.bfloat16, dev).normal_()
    ein = ops.new_nhwc(B, C, H, W, torch.bfloat16, dev)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.1
    w_tnc, w_tcn = torch.empty(9, C, C, device=dev), torch.empty(9, C, C, device=dev)
    st = N.stream()
    N.call('tss_permute_w3x3', N.ptr(w), N.ptr(w_tnc), N.ptr(w_tcn), C, C, st)
    stats = torch.empty(S, 2 * C, dtype=torch.float64, device=dev)
    dw = torch.zeros(C, C, 3, 3, device=dev)

    def fwd():
        N.call('tss_conv3x3_fwd', N.ptr(x), C, None, None, None, 0, N.ptr(w_tnc), None, N.ptr(z), 2 * C, N.ptr(stats), B, H, W, C, C, 2, 1, BF, st)

    def bwd():
        N.call('tss_convkxk_bwd_data', N.ptr(e), 2 * C, None, 0, None, None, None, None, N.ptr(w_tcn), None, 0, None, None, None, 0,
               N.ptr(ein), C, None, B, H, W, C, C, 3, 3, 2, 1, BF, st)

    for tag, fn, out, alg in (('fwd', fwd, lambda: (z[:, :C].clone(), stats.sum(0)), (P + Po) * C * 2),
                              ('bwd', bwd, lambda: (ein.clone(),), (P + Po) * C * 2)):
        fn(); torch.cuda.synchronize(); lean = out(); t1 = timeit(fn)
        N.call('tss_set_option', 1, 1)
        try:
            fn(); torch.cuda.synchronize(); gen = out(); t0 = timeit(fn, 3)
        finally:
            N.call('tss_set_option', 1, 0)
        print('%-5s %dx%dx%d C=%d  lean %7.1f us (%5.0f GB/s)  generic %7.1f us   rel %s' % (
            tag, B, H, W, C, t1, alg / t1 / 1e3, t0, ' '.join('%.2e' % rl(a, b) for a, b in zip(lean, gen))), flush=True)
    rows = N.lib().tss_sconv_bwd_weight_rows(B, H, W, C, C, BF)
    wsr = torch.empty(max(rows, 1), 9 * C * C, device=dev)

    def wg_sweep():
        N.call('tss_sconv_bwd_weight_sweep', N.ptr(e), 2 * C, None, 0, None, None, None, None, N.ptr(x), C, None, None, None, 0, N.ptr(wsr),
               B, H, W, C, C, BF, st)
        ops._reduce_rows_now(wsr, dw, 9 * C * C, rows)

    def wg_generic():
        N.call('tss_conv3x3_bwd_weight', N.ptr(e), 2 * C, None, 0, None, None, None, None, N.ptr(x), C, None, None, None, 0, N.ptr(dw),
               B, H, W, C, C, 2, 1, BF, st)
    dw.zero_(); wg_sweep(); torch.cuda.synchronize(); d1 = dw.clone(); t1 = timeit(wg_sweep)
    dw.zero_(); wg_generic(); torch.cuda.synchronize(); d0 = dw.clone(); t0 = timeit(wg_generic, 3)
    print('%-5s %dx%dx%d C=%d  one sweep %7.1f us (%5.0f GB/s)  generic %7.1f us  rows %d  rel %.2e' % (
        'wgrad', B, H, W, C, t1, (P + Po) * C * 2 / t1 / 1e3, t0, rows, rl(d1, d0)), flush=True)
