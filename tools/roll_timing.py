#!/usr/bin/env python
"""Phase timers of dw_bwd_roll_s1_kernel (debug variant built by tools/ab_variants.sh dwroll.hip timing "-DTSS_ROLL_TIMING";
run with TSS_HIP_LIB=.../variants/libtss_hip_timing.so): cycles of wave 0 per block, averaged."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_semantic_segmentation_amd import _native as N, ops

dev = 'cuda:0'
S = N.stat_slabs()
lib = ctypes.CDLL(N.LIB_PATH)
buf = (ctypes.c_ulonglong * 8)()
for (B, C, H, W) in ((8, 384, 64, 128), (8, 768, 32, 64), (8, 128, 128, 256)):
    mk = lambda: ops.new_nhwc(B, C, H, W, torch.bfloat16, dev).normal_()
    x, e, y, ei = mk(), mk(), mk(), mk()
    v = lambda s=0.1: torch.randn(C, device=dev) * s
    ga, gb, gce, gmu, mean, sc, bias = torch.rand(C, device=dev) + 0.5, v(0.05), v(0.01), v(), v(), torch.rand(C, device=dev) + 0.5, v()
    w = torch.randn(C, 9, device=dev) * 0.3
    ws = torch.empty(S, C * 9, device=dev); dwt = torch.zeros(C, 9, device=dev); bst = torch.empty(S, 2 * C, dtype=torch.float64, device=dev)
    st = N.stream()
    def run():
        N.call('tss_dwconv3x3_bwd_fused', N.ptr(e), C, N.ptr(y), C, N.ptr(ga), N.ptr(gb), N.ptr(gce), N.ptr(gmu), N.ptr(w), N.ptr(x), C,
               N.ptr(mean), N.ptr(sc), N.ptr(bias), 1, 1, N.ptr(ei), C, N.ptr(bst), N.ptr(ws), N.ptr(dwt), B, H, W, C, 1, 1, N.TSS_BF16, st)
    for _ in range(3): run()
    torch.cuda.synchronize()
    lib.tss_debug_roll_timing(buf, 1)
    n = 5
    for _ in range(n): run()
    torch.cuda.synchronize()
    lib.tss_debug_roll_timing(buf, 1)
    blocks = buf[7] / n
    names = ['wait row', 'address + request', 'barrier', 'window+sums+emit', 'prologue', 'tail', 'transform+park']
    tot = sum(buf[i] for i in range(7))
    print('%dx%dx%dx%d: %d blocks; cycles of wave 0 per block (100 MHz s_memtime ticks): ' % (B, C, H, W, blocks)
          + ', '.join('%s %.0f (%.0f%%)' % (names[i], buf[i] / buf[7], 100.0 * buf[i] / tot) for i in range(7)))
