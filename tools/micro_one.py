#!/usr/bin/env python
"""Run ONE kernel configuration repeatedly (for rocprofv3 --pmc): micro_one.py <pwfwd|pwbwd|wgrad|dwfwd> K N P"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_semantic_segmentation_amd import _native as N, ops
which, K, Nn, P = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = 'cuda:0'; S = N.stat_slabs()
for kv in os.environ.get('TSS_OPT', '').split(','):   # e.g. TSS_OPT=2=0 -> tss_set_option(2, 0)
    if kv:
        N.call('tss_set_option', int(kv.split('=')[0]), int(kv.split('=')[1]))
x = torch.randn(P, K, device=dev).bfloat16(); y = torch.empty(P, Nn, device=dev, dtype=torch.bfloat16)
e = torch.randn(P, Nn, device=dev).bfloat16(); ein = torch.empty(P, K, device=dev, dtype=torch.bfloat16)
w = torch.randn(Nn, K, device=dev) * 0.1; dw = torch.zeros(Nn, K, device=dev)
stats = torch.empty(S, 2 * Nn, dtype=torch.float64, device=dev); bst = torch.empty(S, 2 * K, dtype=torch.float64, device=dev)
mK = torch.zeros(K, device=dev); sK = torch.ones(K, device=dev); mN = torch.zeros(Nn, device=dev); sN = torch.ones(Nn, device=dev)
st = N.stream()
nws = N.lib().tss_pwconv_bwd_weight_ws(P, K, Nn, 1) if which == 'wgrad' else 0
wsw = torch.empty(nws, device=dev) if nws and os.environ.get('TSS_WG_ATOMIC') != '1' else None
fns = {
 'pwfwd': lambda: N.call('tss_pwconv_fwd', N.ptr(x), K, N.ptr(mK), N.ptr(sK), N.ptr(mK), 1, N.ptr(w), None, None, N.ptr(y), Nn, N.ptr(stats), P, K, Nn, 1, st),
 'pwbwd': lambda: N.call('tss_pwconv_bwd_data', N.ptr(e), Nn, N.ptr(y), Nn, N.ptr(sN), N.ptr(sN), N.ptr(mN), N.ptr(mN), N.ptr(w), None, N.ptr(x), K, N.ptr(mK), N.ptr(sK), N.ptr(mK), 1, N.ptr(ein), K, N.ptr(bst), None, None, 0, 0, 0, P, K, Nn, 1, st),
 'wgrad': lambda: N.call('tss_pwconv_bwd_weight', N.ptr(e), Nn, N.ptr(y), Nn, N.ptr(sN), N.ptr(sN), N.ptr(mN), N.ptr(mN), N.ptr(x), K, N.ptr(mK), N.ptr(sK), N.ptr(mK), 1, N.ptr(dw), N.ptr(wsw), 0, P, K, Nn, 1, None, st),
}
if which in ('dwfwd', 'dwbwd', 'dwwg'):
    C, B, H, W = K, 8, Nn, P
    xi = ops.new_nhwc(B, C, H, W, torch.bfloat16, dev); xi.normal_(); yo = ops.new_nhwc(B, C, H, W, torch.bfloat16, dev)
    wd = torch.randn(C, 1, 3, 3, device=dev); sd = torch.empty(S, 2 * C, dtype=torch.float64, device=dev)
    m = torch.zeros(C, device=dev); s1 = torch.ones(C, device=dev)
    fns['dwfwd'] = lambda: N.call('tss_dwconv3x3_fwd', N.ptr(xi), C, N.ptr(m), N.ptr(s1), N.ptr(m), 1, N.ptr(wd), N.ptr(yo), C, N.ptr(sd), B, H, W, C, 1, 1, 1, st)
    ei = ops.new_nhwc(B, C, H, W, torch.bfloat16, dev); ei.normal_(); eo = ops.new_nhwc(B, C, H, W, torch.bfloat16, dev)
    dwd = torch.zeros(C, 1, 3, 3, device=dev); ws = torch.empty(S, C * 9, device=dev)
    fns['dwbwd'] = lambda: N.call('tss_dwconv3x3_bwd_data', N.ptr(ei), C, N.ptr(yo), C, N.ptr(s1), N.ptr(s1), N.ptr(m), N.ptr(m), N.ptr(wd), N.ptr(xi), C, N.ptr(m), N.ptr(s1), N.ptr(m), 1, N.ptr(eo), C, N.ptr(sd), None, None, B, H, W, C, 1, 1, 1, st)
    fns['dwwg'] = lambda: N.call('tss_dwconv3x3_bwd_weight', N.ptr(ei), C, N.ptr(yo), C, N.ptr(s1), N.ptr(s1), N.ptr(m), N.ptr(m), N.ptr(xi), C, N.ptr(m), N.ptr(s1), N.ptr(m), 1, N.ptr(dwd), N.ptr(ws), 0, B, H, W, C, 1, 1, 1, st)
if which == 'ceup':
    B, C, h, w, sc = 8, K, Nn, P, 8
    low = ops.new_nhwc(B, C, h, w, torch.bfloat16, dev); low.normal_()
    tgt = torch.randint(0, C, (B, h * sc, w * sc), device=dev)
    ws = torch.empty(N.lib().tss_upsample_ce_ws(B, C, h, w, h * sc, w * sc), dtype=torch.float32, device=dev)
    scal = torch.empty(2, dtype=torch.float32, device=dev)
    fns['ceup'] = lambda: N.call('tss_upsample_ce_fwd', N.ptr(low), low.stride(3), N.ptr(tgt), N.ptr(ws), N.ptr(scal[0:1]), N.ptr(scal[1:2]), B, C, h, w, h * sc, w * sc, 255, 1, st)
for _ in range(10):
    fns[which]()
torch.cuda.synchronize()
if os.environ.get('TSS_TIMING') == '1' and which.startswith('pw'):
    import ctypes
    buf = (ctypes.c_ulonglong * 8)()
    N.lib().tss_debug_pw_timing(buf, 1)
    n = max(buf[7], 1)
    for q, nm in enumerate(['prologue', 'loop', 'tail', ' pro:setup', ' pro:issue', ' pro:weights', ' pro:consts']):
        print('%-12s %10.0f cycles/block' % (nm, buf[q] / n))
    print('blocks', n // 10)
elif os.environ.get('TSS_TIMING') == '1' and which.startswith('dw'):
    import ctypes
    buf = (ctypes.c_ulonglong * 8)()
    N.lib().tss_debug_dw_timing(buf, 1)
    n = max(buf[7], 1)
    for q, nm in enumerate(['prologue', 'loop', 'tail']):
        print('%-10s %10.0f cycles/block' % (nm, buf[q] / n))
    print('blocks', n // 10)
elif os.environ.get('TSS_TIMING') == '1':
    import ctypes
    buf = (ctypes.c_ulonglong * 8)()
    N.lib().tss_debug_wg_timing(buf, 1)
    n = max(buf[7], 1)
    names = ['barrier_in', 'G_half(wait+xform)', 'A_half', 'barrier_out', 'mfma', 'prologue', 'atomics_tail']
    tot = sum(buf[q] for q in range(7))
    for q in range(7):
        print('%-20s %10.0f cycles/block  %5.1f %%' % (names[q], buf[q] / n, 100.0 * buf[q] / max(tot, 1)))
    print('blocks', n // 10)
