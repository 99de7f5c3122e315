"""Stand-alone timing of the dense 3x3 128 -> 128 kernels on the ASPP map of BASELINE config 5 (1 x 128 x 256 x 512, rates 6 / 12 / 18):
csrc/wstat.hip (weights stationary in registers; default) or, with TSS_CONV3X3_WSTAT=0, csrc/atrous.hip (activations streamed).
usage: python tools/micro_atrous.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_semantic_segmentation_amd import _native as N, ops
dev = 'cuda:0'
B, C, H, W = 1, 128, 256, 512
xs = [ops.new_nhwc(B, C, H, W, torch.bfloat16, dev) for _ in range(4)]
for x in xs: x.copy_(torch.randn(B, C, H, W, device=dev))
w = torch.randn(C, C, 3, 3, device=dev) * 0.05
w16 = torch.empty((9, C, C), dtype=torch.bfloat16, device=dev)
st = N.stream()
N.call('tss_permute_w3x3_bf16', N.ptr(w), N.ptr(w16), None, C, C, st)
ys = [ops.new_nhwc(B, C, H, W, torch.bfloat16, dev) for _ in range(4)]
flops = 2.0 * B * H * W * C * C * 9
which = 'atrous.hip stream' if os.environ.get('TSS_CONV3X3_WSTAT') == '0' else 'wstat.hip'
for dbg in [which]:
    for dil in (6, 12, 18):
        def run(i):
            N.call('tss_conv3x3_fwd', N.ptr(xs[i % 4]), xs[0].stride(3), None, None, None, 0, None, N.ptr(w16), N.ptr(ys[i % 4]), ys[0].stride(3), None, B, H, W, C, C, 1, dil, 1, st)
        for i in range(5): run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(40): run(i)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 40
        print('%s dil %2d: %7.1f us  %6.1f TFLOP/s  (%.1f %% of 2.5 PFLOP/s)' % (dbg, dil, us, flops / us / 1e6, flops / us / 1e6 / 25))
