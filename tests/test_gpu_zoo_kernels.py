"""GPU parity of the round-4 kernels of the LEDNet / ESNet rows THROUGH THE C ABI, against the generic
implicit-GEMM kernels on the same operands (tss_set_option(TSS_OPT_DISABLE_FAST_PATHS, 1)) -- ragged maps, pitches wider than the channel
count, every mode (pending BatchNorm + ReLU on load, BatchNorm-backward combination of e and y, ReLU mask + backward statistics) -- and of
the operators built on them against plain torch autograd:
  csrc/fc1d.hip   three-tap 1x3 / 3x1 layers, 16 / 32 / 64 channels            (TSS/models/lednet.py:157-180)
  csrc/fcg.hip    five taps x 64 channels, three dilated taps x 128 channels   (TSS/models/esnet.py:83-166)
  csrc/sconv.hip  stride-2 3x3 (square), transposed 3x3 and its gradients (rectangular)   (lednet.py:130-131, esnet.py:54-56,71-80)
  csrc/ssnbt.hip  the tail of the SS-nbt unit, ops.split_fork, ops.channel_slice          (lednet.py:112-124)
The f64-oracle comparison of the same kernels lives in tests/test_gpu_lean_vs_oracle.py; this file pins the C entry points themselves."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def rl(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30)).item()


def both(fn):
    """(lean results, generic results) of the same C calls"""
    from torch_semantic_segmentation_amd import _native as N
    fn()
    torch.cuda.synchronize()
    lean = fn()
    torch.cuda.synchronize()
    N.call('tss_set_option', 1, 1)
    try:
        gen = fn()
        torch.cuda.synchronize()
    finally:
        N.call('tss_set_option', 1, 0)
    return lean, gen


TAP_CASES = [   # (C, taps, B, H, W, axis, dilation)
    (16, 3, 2, 13, 37, 0, 1), (16, 3, 2, 13, 37, 1, 2), (32, 3, 3, 9, 70, 0, 2), (32, 3, 1, 22, 18, 1, 5), (64, 3, 2, 11, 21, 0, 9),
    (64, 3, 2, 19, 16, 1, 17), (64, 5, 2, 11, 27, 0, 1), (64, 5, 1, 21, 33, 1, 1), (128, 3, 2, 9, 20, 0, 2), (128, 3, 1, 23, 17, 1, 9),
]


@pytest.mark.parametrize('C,T,B,H,W,axis,dil', TAP_CASES)
def test_factorized_tap_kernels_match_the_generic_kernels(C, T, B, H, W, axis, dil):
    from torch_semantic_segmentation_amd import _native as N, ops
    torch.manual_seed(C + T + axis)
    BF, S, st = N.TSS_BF16, N.stat_slabs(), N.stream()
    ldx, lde = C + 8, C + 16
    P = B * H * W
    buf = lambda ld_: torch.randn(P, ld_, device=DEV).to(torch.bfloat16)
    x, e, y = buf(ldx), buf(lde), buf(lde)
    v = lambda s_=0.1: torch.randn(C, device=DEV) * s_
    mean, sc, bias, cb = v(), torch.rand(C, device=DEV) + 0.5, v(), v()
    ga, gb, gce, gmu = torch.rand(C, device=DEV) + 0.5, v(0.05), v(0.01), v()
    w = torch.randn(C, C, T, device=DEV) * 0.15
    w_tnc, w_tcn = torch.empty(T, C, C, device=DEV), torch.empty(T, C, C, device=DEV)
    N.call('tss_permute_wtaps', N.ptr(w), N.ptr(w_tnc), N.ptr(w_tcn), C, C, T, st)
    kh, kw = (1, T) if axis == 0 else (T, 1)

    def fwd():
        out = torch.zeros(P, lde, device=DEV).to(torch.bfloat16)
        stats = torch.empty(S, 2 * C, dtype=torch.float64, device=DEV)
        if T == 3:
            N.call('tss_conv1d3_fwd', N.ptr(x), ldx, N.ptr(mean), N.ptr(sc), N.ptr(bias), 1, N.ptr(w_tnc), N.ptr(cb), N.ptr(out), lde,
                   N.ptr(stats), B, H, W, C, C, axis, dil, BF, st)
        else:
            N.call('tss_convkxk_fwd', N.ptr(x), ldx, N.ptr(mean), N.ptr(sc), N.ptr(bias), 1, N.ptr(w_tnc), N.ptr(cb), N.ptr(out), lde,
                   N.ptr(stats), B, H, W, C, C, kh, kw, 1, dil, BF, st)
        return out.float(), stats.sum(0)

    def bwd(with_y):
        ein = torch.zeros(P, ldx, device=DEV).to(torch.bfloat16)
        bst = torch.empty(S, 2 * C, dtype=torch.float64, device=DEV)
        gargs = ((N.ptr(e), lde, N.ptr(y), lde, N.ptr(ga), N.ptr(gb), N.ptr(gce), N.ptr(gmu)) if with_y
                 else (N.ptr(e), lde, None, 0, None, None, None, None))
        if T == 3:
            N.call('tss_conv1d3_bwd_data', *gargs, N.ptr(w_tcn), N.ptr(x), ldx, N.ptr(mean), N.ptr(sc), N.ptr(bias), 1, N.ptr(ein), ldx,
                   N.ptr(bst), B, H, W, C, C, axis, dil, BF, st)
        else:
            N.call('tss_convkxk_bwd_data', *gargs, N.ptr(w_tcn), N.ptr(x), ldx, N.ptr(mean), N.ptr(sc), N.ptr(bias), 1, N.ptr(ein), ldx,
                   N.ptr(bst), B, H, W, C, C, kh, kw, 1, dil, BF, st)
        return ein.float(), bst.sum(0)

    (o1, s1), (o0, s0) = both(fwd)
    assert rl(o1, o0) < 2e-3 and rl(s1, s0) < 1e-4, ('fwd', rl(o1, o0), rl(s1, s0))       # bf16 outputs: the rare 1-ulp rounding difference
    assert torch.equal(o1[:, C:], o0[:, C:])                                              # nothing written beyond the channel count
    for with_y in (True, False):
        (g1, b1), (g0, b0) = both(lambda: bwd(with_y))
        assert rl(g1, g0) < 2e-3 and rl(b1, b0) < 2e-3, ('bwd', with_y, rl(g1, g0), rl(b1, b0))
        assert torch.equal(g1[:, C:], g0[:, C:])

    # weight gradient: one sweep + row reduction against the generic tap-loop kernel
    rows = (N.lib().tss_conv1d3_bwd_weight_rows(P, C, C, BF) or N.lib().tss_convtap_bwd_weight_rows(P, C, C, 3, BF)) if T == 3 \
        else N.lib().tss_convtap_bwd_weight_rows(P, C, C, T, BF)
    assert rows > 0
    for with_y in (True, False):
        gargs = ((N.ptr(e), lde, N.ptr(y), lde, N.ptr(ga), N.ptr(gb), N.ptr(gce), N.ptr(gmu)) if with_y
                 else (N.ptr(e), lde, None, 0, None, None, None, None))
        xargs = (N.ptr(x), ldx, N.ptr(mean), N.ptr(sc), N.ptr(bias), 1)
        ws = torch.full((rows, C * C * T), float('nan'), device=DEV)
        dw1 = torch.zeros(C, C, T, device=DEV)
        if T == 3 and C <= 64:
            N.call('tss_conv1d3_bwd_weight_sweep', *gargs, *xargs, N.ptr(ws), B, H, W, C, C, axis, dil, BF, st)
        else:
            N.call('tss_convtap_bwd_weight_sweep', *gargs, *xargs, N.ptr(ws), B, H, W, C, C, T, axis, dil, BF, st)
        ops._reduce_rows_now(ws, dw1, C * C * T, rows)
        dw0 = torch.zeros(C, C, T, device=DEV)
        N.call('tss_set_option', 1, 1)
        try:
            if T == 3:
                N.call('tss_conv1d3_bwd_weight', *gargs, *xargs, N.ptr(dw0), B, H, W, C, C, axis, dil, BF, st)
            else:
                N.call('tss_convkxk_bwd_weight', *gargs, *xargs, N.ptr(dw0), B, H, W, C, C, kh, kw, 1, dil, BF, st)
        finally:
            N.call('tss_set_option', 1, 0)
        torch.cuda.synchronize()
        assert rl(dw1, dw0) < 1e-4, ('dw', with_y, rl(dw1, dw0))


@pytest.mark.parametrize('B,H,W,C', [(2, 38, 50, 32), (3, 22, 36, 64), (1, 6, 4, 32)])
def test_strided_3x3_kernels_match_the_generic_kernels(B, H, W, C):
    from torch_semantic_segmentation_amd import _native as N, ops
    torch.manual_seed(C)
    BF, S, st = N.TSS_BF16, N.stat_slabs(), N.stream()
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = ops.new_nhwc(B, C, H, W, torch.bfloat16, DEV).normal_()
    e = ops.new_nhwc(B, 2 * C, Ho, Wo, torch.bfloat16, DEV).normal_()        # the convolution owns channels [0, C) of the concat buffer
    v = lambda s_=0.1: torch.randn(C, device=DEV) * s_
    mean, sc, bias, cb = v(), torch.rand(C, device=DEV) + 0.5, v(), v()
    w = torch.randn(C, C, 3, 3, device=DEV) * 0.1
    w_tnc, w_tcn = torch.empty(9, C, C, device=DEV), torch.empty(9, C, C, device=DEV)
    N.call('tss_permute_w3x3', N.ptr(w), N.ptr(w_tnc), N.ptr(w_tcn), C, C, st)

    def fwd():
        z = ops.new_nhwc(B, 2 * C, Ho, Wo, torch.bfloat16, DEV).zero_()
        stats = torch.empty(S, 2 * C, dtype=torch.float64, device=DEV)
        N.call('tss_convkxk_fwd', N.ptr(x), C, N.ptr(mean), N.ptr(sc), N.ptr(bias), 1, N.ptr(w_tnc), N.ptr(cb), N.ptr(z), 2 * C, N.ptr(stats),
               B, H, W, C, C, 3, 3, 2, 1, BF, st)
        return z.float(), stats.sum(0)

    def bwd():
        ein = ops.new_nhwc(B, C, H, W, torch.bfloat16, DEV).zero_()
        N.call('tss_convkxk_bwd_data', N.ptr(e), 2 * C, None, 0, None, None, None, None, N.ptr(w_tcn), None, 0, None, None, None, 0,
               N.ptr(ein), C, None, B, H, W, C, C, 3, 3, 2, 1, BF, st)
        return (ein.float(),)

    (z1, s1), (z0, s0) = both(fwd)
    assert rl(z1[:, :C], z0[:, :C]) < 2e-3 and rl(s1, s0) < 1e-4 and torch.equal(z1[:, C:], z0[:, C:])
    (g1,), (g0,) = both(bwd)
    assert rl(g1, g0) < 2e-3
    rows = N.lib().tss_sconv_bwd_weight_rows(B, H, W, C, C, BF)
    assert rows > 0
    ws = torch.full((rows, 9 * C * C), float('nan'), device=DEV)
    dw1, dw0 = torch.zeros(C, C, 3, 3, device=DEV), torch.zeros(C, C, 3, 3, device=DEV)
    xargs = (N.ptr(x), C, N.ptr(mean), N.ptr(sc), N.ptr(bias), 1)
    N.call('tss_sconv_bwd_weight_sweep', N.ptr(e), 2 * C, None, 0, None, None, None, None, *xargs, N.ptr(ws), B, H, W, C, C, BF, st)
    ops._reduce_rows_now(ws, dw1, 9 * C * C, rows)
    N.call('tss_conv3x3_bwd_weight', N.ptr(e), 2 * C, None, 0, None, None, None, None, *xargs, N.ptr(dw0), B, H, W, C, C, 2, 1, BF, st)
    torch.cuda.synchronize()
    assert rl(dw1, dw0) < 1e-4


@pytest.mark.parametrize('ci,co,B,h,w', [(64, 16, 2, 11, 19), (16, 24, 2, 13, 20), (16, 16, 1, 5, 7), (128, 64, 2, 9, 13)])
def test_transposed_3x3_operator_matches_torch(ci, co, B, h, w):
    """ops.conv_transpose (ConvTranspose2d(3, stride 2, padding 1, output_padding 1) + bias, TSS/models/esnet.py:71-80) on the rectangular
    instances of csrc/sconv.hip: forward, input gradient and weight / bias gradients against torch autograd on the same bf16 operands."""
    from torch_semantic_segmentation_amd import ops
    torch.manual_seed(ci + co)
    x = (torch.randn(B, ci, h, w, device=DEV)).to(torch.bfloat16)
    wt = (torch.randn(ci, co, 3, 3, device=DEV) * 0.2).to(torch.bfloat16).float().requires_grad_(True)
    bias = torch.randn(co, device=DEV).requires_grad_(True)
    xs = ops.to_nhwc(x).requires_grad_(True)
    y = ops.conv_transpose(xs, wt, bias, 2)
    cot = torch.randn(B, co, 2 * h, 2 * w, device=DEV).to(torch.bfloat16)
    y.backward(ops.to_nhwc(cot))
    xr = x.float().requires_grad_(True)
    wr, br = wt.detach().clone().requires_grad_(True), bias.detach().clone().requires_grad_(True)
    yr = torch.nn.functional.conv_transpose2d(xr, wr, br, stride=2, padding=1, output_padding=1)
    yr.backward(cot.float())
    assert rl(y, yr) < 4e-3                                   # one bf16 rounding of the output
    assert rl(xs.grad, xr.grad) < 4e-3
    assert rl(wt.grad, wr.grad) < 2e-3 and rl(bias.grad, br.grad) < 2e-3


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('C,p', [(32, 0.0), (64, 0.3), (128, 0.3)])
def test_ssnbt_tail_in_one_pass_matches_the_four_operators(C, p, dtype):
    """ops.ssnbt_tail (csrc/ssnbt.hip) against concat_joined -> channel_dropout -> join -> channel_shuffle of the same library, with the
    same dropout draw: outputs, the gradients of both branches' raw tensors and of the skip, and both branches' BatchNorm-backward sums."""
    from torch_semantic_segmentation_amd import ops
    B, H, W = 2, 9, 14
    half = C // 2
    results = []
    for fused in (True, False):
        torch.manual_seed(3)
        bnl, bnr = torch.nn.BatchNorm2d(half).to(DEV), torch.nn.BatchNorm2d(half).to(DEV)
        convl = torch.nn.Conv2d(half, half, 1, bias=False).to(DEV)
        convr = torch.nn.Conv2d(half, half, 1, bias=False).to(DEV)
        with torch.no_grad():
            for bn in (bnl, bnr):
                bn.weight.uniform_(0.6, 1.4); bn.bias.uniform_(-0.3, 0.3)
        x = ops.to_nhwc(torch.randn(B, C, H, W, device=DEV).to(dtype)).requires_grad_(True)
        xl, xr, xs = ops.split_fork(x)
        left = ops.conv_unit(xl, convl, bnl, False)
        right = ops.conv_unit(xr, convr, bnr, False)
        old = ops.fuse_ssnbt_tail
        ops.fuse_ssnbt_tail = fused
        try:
            torch.manual_seed(11)                      # the dropout draw
            out = ops.ssnbt_tail(left, right, xs, p, True)
        finally:
            ops.fuse_ssnbt_tail = old
        out.backward(ops.to_nhwc(torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(5)).to(DEV).to(dtype)))
        torch.cuda.synchronize()
        results.append((out.detach().float(), x.grad.float(), convl.weight.grad.clone(), convr.weight.grad.clone(),
                        bnl.weight.grad.clone(), bnr.bias.grad.clone()))
    # f32 pins every tensor at 1e-5.  In bf16 the unfused path rounds three intermediate tensors, so ~0.4 % of the units decide their ReLU
    # differently; each flip moves a gradient element by its full magnitude: sqrt(0.004) = 6 % relative L2 on every gradient, whatever the map
    # size (tests/test_gpu_lean_vs_oracle.py holds the fused kernels to the f64 oracle under the noise model instead)
    for i, (a, b) in enumerate(zip(*results)):
        tol = 1e-5 if dtype == torch.float32 else (2e-2 if i == 0 else (0.15 if i < 4 else 0.3))
        assert rl(a, b) < tol, (i, rl(a, b))
    assert not ops._passes


def test_split_fork_and_channel_slice_gradients_match_autograd():
    from torch_semantic_segmentation_amd import ops
    torch.manual_seed(0)
    x0 = torch.randn(2, 32, 5, 7, device=DEV)
    x = ops.to_nhwc(x0.clone()).requires_grad_(True)
    a, b, s = ops.split_fork(x)
    ((a * 2).sum() + (b * 3).sum() + torch.relu(s).sum()).backward()
    xr = x0.clone().requires_grad_(True)
    l, r = torch.chunk(xr, 2, 1)
    ((l * 2).sum() + (r * 3).sum() + torch.relu(xr).sum()).backward()
    assert torch.equal(x.grad, xr.grad)
    for dtype in (torch.float32, torch.bfloat16):
        z = ops.new_nhwc(2, 24, 6, 5, dtype, DEV).normal_().requires_grad_(True)
        y = ops.channel_slice(z, 19)
        assert y.shape == (2, 19, 6, 5) and y.data_ptr() == z.data_ptr()
        g = torch.randn(2, 19, 6, 5, device=DEV).to(dtype)
        y.backward(g)
        assert torch.equal(z.grad[:, :19], g) and not z.grad[:, 19:].any()
