"""GPU parity of the SURVEY.md section 8f N4 rows: the PSPNet head (TSS/models/pspnet.py) against golden vectors generated from
the reference itself, and the DeepLab-style ASPP head of BASELINE config 5 against plain torch (the reference has no ASPP:
its parity is pinned to torch.nn.functional, oracle/aspp.py says so)."""
import os

import numpy as np
import pytest
import torch
from torch import nn

from oracle import aspp as OA
from oracle.recipe import formula_state, lattice_input, lattice_target, synthetic_batch
from tests import cases

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def close(a, b, rel=1e-3, floor=2e-4):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() <= rel * np.abs(b).max() + floor


@pytest.mark.parametrize('mode', ['eval', 'train'])
@pytest.mark.parametrize('name', sorted(cases.PSP_SHAPES))
def test_pspnet_head_matches_reference(golden_dir, name, mode):
    import torch_semantic_segmentation_amd as tssa
    g = cases.load_npz(os.path.join(golden_dir, 'pspnet.npz'))
    m = cases.product_psp(name)
    m.load_state_dict(formula_state(m), strict=True)
    m.train(mode == 'train').to(DEV)
    tssa.set_compute_dtype(m, torch.float32)
    x = cases.psp_inputs(name)[0].to(DEV).requires_grad_(True)
    out = m(x)
    out.backward(cases.block_cotangent(out.shape).to(DEV))
    key = '%s/%s/' % (mode, name)
    assert tuple(out.shape) == g[key + 'out'].shape
    assert close(out.detach().cpu().numpy(), g[key + 'out']), 'forward'
    assert close(x.grad.cpu().numpy(), g[key + 'dx0']), 'dx'
    for pname, p in m.named_parameters():
        # the 1x1-bin arm normalises over B values per channel (xhat = +-1 for B = 2): looser, as for FastSCNN's pyramid
        rel = 5e-3 if (mode == 'train' and name.startswith('psp') and (pname.startswith('0.') or pname.startswith('ppm.0.'))) else 1e-3
        assert close(p.grad.cpu().numpy(), g[key + 'dw.' + pname], rel=rel), pname


def test_lednet_unit_public_surface_and_shuffle():
    """SSnbtBlock / FactorizedConvBlock keep the reference's state_dict keys and ValueErrors (TSS/models/lednet.py:99-100,158-159);
    channel_shuffle (lednet.py:183-188) is bit-exact and its backward is the inverse permutation."""
    from torch_semantic_segmentation_amd.models import lednet as L
    m, o = L.SSnbtBlock(64, 64, dilation=2), OA.SSnbt(64, 2)
    assert list(m.state_dict()) == list(o.state_dict())
    with pytest.raises(ValueError):
        L.SSnbtBlock(64, 32)
    with pytest.raises(ValueError):
        L.FactorizedConvBlock(32, 16)
    x = torch.randn(2, 48, 5, 7, device=DEV, requires_grad=True)
    for groups in (2, 3, 6):
        y = L.channel_shuffle(x, groups)
        assert torch.equal(y.detach().cpu(), OA.shuffle(x.detach().cpu(), groups))
        g = torch.randn_like(y)
        (gx,) = torch.autograd.grad(y, x, g)
        ref = x.detach().cpu().clone().requires_grad_(True)
        OA.shuffle(ref, groups).backward(g.cpu())
        assert torch.equal(gx.cpu(), ref.grad)


def test_pspnet_state_dict_and_hooks():
    from torch_semantic_segmentation_amd.models.pspnet import PSPNet
    m = PSPNet(nn.Identity(), 19, 64)
    o = OA.PSPNetOracle(nn.Identity(), 19, 64)
    assert list(m.state_dict()) == list(o.state_dict())
    m.load_state_dict(formula_state(o), strict=True)
    m.to(DEV).eval()
    seen = {}
    h = m.ppm.register_forward_hook(lambda mod, i, out: seen.__setitem__('pools', out))
    x = lattice_input(2, 64, 12, 20).to(DEV)
    with torch.no_grad():
        y_hooked = m(x)
    h.remove()
    with torch.no_grad():
        y = m(x)
    assert tuple(seen['pools'].shape) == (2, 64, 12, 20)          # the pools alone, as the reference's module returns
    assert cases.rel_err(y_hooked.cpu().numpy(), y.cpu().numpy()) < 1e-6


@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_aspp_head_matches_torch(mode):
    """ASPPHead(64 -> 19, rates 2/4/6, mid 64) on a 2 x 64 x 24 x 40 map, f32: forward, dX and every parameter gradient against
    the plain-torch module (pinned to torch, not to the reference: it has no ASPP)."""
    import torch_semantic_segmentation_amd as tssa
    from torch_semantic_segmentation_amd.models.aspp import ASPPHead
    torch.manual_seed(0)
    ref = OA.ASPPHead(64, 19, atrous_rates=(2, 4, 6), mid_channels=64)
    ref.load_state_dict(formula_state(ref), strict=True)
    cases.zero_dropout(ref)
    m = ASPPHead(64, 19, atrous_rates=(2, 4, 6), mid_channels=64)
    m.load_state_dict(ref.state_dict(), strict=True)
    cases.zero_dropout(m)
    ref.train(mode == 'train'); m.train(mode == 'train').to(DEV)
    tssa.set_compute_dtype(m, torch.float32)
    x0 = lattice_input(2, 64, 24, 40)
    xr = x0.clone().requires_grad_(True)
    out_r = ref(xr)
    cot = cases.block_cotangent(out_r.shape)
    out_r.backward(cot)
    x = x0.clone().to(DEV).requires_grad_(True)
    out = m(x)
    out.backward(cot.to(DEV))
    assert close(out.detach().cpu().numpy(), out_r.detach().numpy()), 'forward'
    assert close(x.grad.cpu().numpy(), xr.grad.numpy(), floor=5e-4), 'dx'
    for (pn, p), (_, r) in zip(m.named_parameters(), ref.named_parameters()):
        # the image-pooling branch normalises over B = 2 values per channel in train mode: looser
        rel = 1e-2 if (mode == 'train' and '.convs.4.' in '.' + pn) else 2e-3
        assert close(p.grad.cpu().numpy(), r.grad.numpy(), rel=rel, floor=2e-3), pn


def test_fastscnn_aspp_eval_logits_and_argmax_vs_torch():
    """The config-5 stress model (FastSCNN trunk + ASPP head + x8 upsample), eval mode, seeded N(0,1) input at 2 x 3 x 256 x 512:
    logits within 1e-3, argmax exact outside sub-resolution ties; bf16 runs and stays within bf16 noise of f32."""
    import torch_semantic_segmentation_amd as tssa
    from torch_semantic_segmentation_amd.models.aspp import fastscnn_aspp
    torch.manual_seed(0)
    ref = OA.FastSCNNASPP(3, 19)
    m = fastscnn_aspp(3, 19)
    m.load_state_dict(ref.state_dict(), strict=True)
    ref.eval(); m.eval().to(DEV)
    x, _ = synthetic_batch(2, 256, 512)
    with torch.no_grad():
        want = ref(x)
        got = m(x.to(DEV))
        pred, _ = tssa.argmax_confusion(got)
    rel = cases.rel_err(got.cpu().numpy(), want.numpy())
    assert rel < 1e-3, rel
    top2 = want.topk(2, dim=1).values
    mism = pred.cpu().long() != want.argmax(1)
    assert ((top2[:, 0] - top2[:, 1])[mism] <= 2 * (got.cpu() - want).abs().max()).all()
    tssa.set_compute_dtype(m, torch.bfloat16)
    with torch.no_grad():
        low16 = m.forward_lowres(x.to(DEV)).float()
        tssa.set_compute_dtype(m, torch.float32)
        low32 = m.forward_lowres(x.to(DEV))
    assert ((low16 - low32).norm() / low32.norm()).item() < 3e-2


def test_fastscnn_aspp_config5_size_vs_oracle():
    """BASELINE config 5 with the DeepLab-style head at its real size, 1 x 3 x 2048 x 4096, eval mode: f32
    logits within 1e-3 of oracle/aspp.py (pinned to torch, not to the reference -- the reference has no ASPP head), argmax
    exact outside sub-resolution ties; the bf16 forward the benchmark times within bf16 noise of it."""
    import torch_semantic_segmentation_amd as tssa
    from torch_semantic_segmentation_amd.models.aspp import fastscnn_aspp
    torch.manual_seed(0)
    ref = OA.FastSCNNASPP(3, 19)
    m = fastscnn_aspp(3, 19)
    m.load_state_dict(ref.state_dict(), strict=True)
    ref.eval(); m.eval().to(DEV)
    x, _ = synthetic_batch(1, 2048, 4096)
    with torch.no_grad():
        want = ref(x)
        got = m(x.to(DEV))
        pred, _ = tssa.argmax_confusion(got)
    rel = cases.rel_err(got.cpu().numpy(), want.numpy())
    top2 = want.topk(2, dim=1).values
    mism = pred.cpu().long() != want.argmax(1)
    err = (got.cpu() - want).abs().max()
    print('fastscnn_aspp 1 x 3 x 2048 x 4096 eval: logits rel err %.2e, argmax mismatch fraction %.2e' % (rel, mism.float().mean().item()))
    assert rel < 1e-3, rel
    assert ((top2[:, 0] - top2[:, 1])[mism] <= 2 * err + 1e-7).all()
    assert mism.float().mean().item() < 1e-4
    del want, got
    tssa.set_compute_dtype(m, torch.bfloat16)
    with torch.no_grad():
        low16 = m.forward_lowres(x.to(DEV)).float()
        tssa.set_compute_dtype(m, torch.float32)
        low32 = m.forward_lowres(x.to(DEV))
    assert ((low16 - low32).norm() / low32.norm()).item() < 3e-2


@pytest.mark.parametrize('mode', ['eval', 'train'])
@pytest.mark.parametrize('name', sorted(cases.ZOO_SHAPES))
def test_lednet_esnet_blocks_match_reference(golden_dir, name, mode):
    """LEDNet's DownsamplingBlock (on the image and on an activation) and APN decoder with 19 classes, ESNet's FCU (K = 3, 5) /
    FPCU blocks and down-sampler: forward, dX and every parameter gradient against vectors of the imported reference
    (tests/golden/zoo.npz; TSS/models/lednet.py:58-92,126-144, TSS/models/esnet.py:47-68,83-166), f32, 1e-3."""
    import torch_semantic_segmentation_amd as tssa
    g = cases.load_npz(os.path.join(golden_dir, 'zoo.npz'))
    m = cases.product_zoo(name)
    o = cases.oracle_zoo(name)
    assert list(m.state_dict()) == list(o.state_dict())
    m.load_state_dict(formula_state(m), strict=True)
    cases.zero_all_dropout(m)
    m.train(mode == 'train').to(DEV)
    tssa.set_compute_dtype(m, torch.float32)
    image = name == 'led_down_img'
    x = cases.zoo_inputs(name)[0].to(DEV).requires_grad_(not image)
    out = m(x)
    out.backward(cases.block_cotangent(out.shape).to(DEV))
    key = '%s/%s/' % (mode, name)
    assert tuple(out.shape) == g[key + 'out'].shape
    assert close(out.detach().cpu().numpy(), g[key + 'out']), 'forward'
    if not image:
        assert close(x.grad.cpu().numpy(), g[key + 'dx0']), 'dx'
    for pname, p in m.named_parameters():
        ref = g[key + 'dw.' + pname]
        if mode == 'train' and (pname.endswith('.2.bias') or pname == 'conv.bias'):
            # a convolution bias in front of a training-mode BatchNorm: the exact gradient is 0 -- the reference returns rounding noise
            # (1e-4 of the layer's weight gradient), the HIP path returns 0
            wref = np.abs(g[key + 'dw.' + pname[:-4] + 'weight']).max()
            assert np.abs(ref).max() < 1e-3 * wref, pname
            assert p.grad is None or p.grad.abs().max().item() < 1e-3 * wref, pname
            continue
        # the level5 arm of the APN normalises over B = 2 values per channel (xhat = +-1): looser, as for the 1x1-bin pyramid arm
        rel = 5e-3 if (mode == 'train' and name == 'led_apn' and pname.startswith('level5')) else 1e-3
        assert close(p.grad.cpu().numpy(), ref, rel=rel), pname
    if mode == 'train':
        for bname, b in m.named_buffers():
            if bname.endswith('running_mean') or bname.endswith('running_var'):
                assert close(b.cpu().numpy(), g[key + 'buf.' + bname]), bname


def test_lednet_whole_model_eval_logits_and_argmax(golden_dir):
    """lednet(3, 19) (TSS/models/lednet.py:13-55) on the 2 x 3 x 64 x 128 lattice image: state_dict keys of the reference, eval-mode
    logits at 1e-3 and the arg-max map (at most 0.1 % of the pixels on the other side of a near-tie) against the imported
    reference's fixture; train-mode logits of the same weights; Dropout2d in training mode zeroes whole channels and keeps the mean."""
    import torch_semantic_segmentation_amd as tssa
    g = cases.load_npz(os.path.join(golden_dir, 'zoo.npz'))
    m = cases.product_zoo('led_net')
    o = cases.oracle_zoo('led_net')
    assert list(m.state_dict()) == list(o.state_dict())
    assert all(tuple(a.shape) == tuple(b.shape) for a, b in zip(m.state_dict().values(), o.state_dict().values()))
    m.load_state_dict(formula_state(m), strict=True)
    cases.zero_all_dropout(m)
    m.to(DEV).eval()
    tssa.set_compute_dtype(m, torch.float32)
    x = lattice_input(*cases.LEDNET_SHAPE).to(DEV)
    with torch.no_grad():
        out = m(x)
        low = m.forward_lowres(x)
    assert tuple(out.shape) == (2, 19, 64, 128)
    assert close(low.float().cpu().numpy(), g['eval/led_net/lowres'])
    assert close(out[:, :, ::4, ::4].float().cpu().numpy(), g['eval/led_net/out_sub4'])
    mism = (out.argmax(1).cpu().numpy().astype(np.uint8) != g['eval/led_net/argmax']).mean()
    assert mism <= 1e-3, mism
    m.train()
    out_t = m(x)
    assert close(out_t[:, :, ::4, ::4].detach().float().cpu().numpy(), g['train/led_net/out_sub4'], rel=5e-3)
    out_t.float().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    for n, b in m.named_buffers():
        if n.endswith('running_mean') or n.endswith('running_var'):
            assert abs(b.double().norm().item() / max(float(g['train/led_net/buf_norm.' + n]), 1e-12) - 1) < 5e-3, n


def test_channel_dropout_zeroes_whole_channels():
    """nn.Dropout2d in training mode (TSS/models/lednet.py:113): per (image, channel) either all zeros or x / (1 - p); backward
    applies the same mask; eval mode and p = 0 are the identity."""
    from torch_semantic_segmentation_amd import ops
    torch.manual_seed(0)
    x = (torch.randn(4, 64, 9, 11, device=DEV) + 3.0).requires_grad_(True)
    y = ops.channel_dropout(x, 0.3, True)
    ratio = (y / x.detach()).detach()
    per = ratio.flatten(2)
    assert torch.allclose(per.min(2)[0], per.max(2)[0])                        # one factor per (image, channel)
    vals = per[:, :, 0]
    assert torch.all((vals == 0) | ((vals - 1 / 0.7).abs() < 1e-5))
    assert 0.1 < (vals == 0).float().mean().item() < 0.5
    y.backward(torch.ones_like(y))
    assert torch.allclose(x.grad.flatten(2)[:, :, 0], vals, atol=1e-6)
    assert ops.channel_dropout(x, 0.3, False) is x and ops.channel_dropout(x, 0.0, True) is x


@pytest.mark.parametrize('use_graph', [False, True])
def test_lednet_trains_under_the_trainer_in_bf16(use_graph):
    """lednet(3, 19) with bf16 activations, Dropout2d active, the fused x8 head + cross-entropy and FlatAdamW, eagerly and as a
    captured HIP graph: finite losses that fall on a fixed batch, BatchNorm running statistics of the padded 19-class layers updated."""
    import torch_semantic_segmentation_amd as tssa
    from torch_semantic_segmentation_amd import engine as E
    torch.manual_seed(0)
    m = cases.product_zoo('led_net').to(DEV)
    tssa.set_compute_dtype(m, torch.bfloat16)
    opt = E.FlatAdamW(m.parameters(), lr=2e-3, weight_decay=1e-5)
    tr = E.Trainer(m, opt, tssa.CrossEntropyLoss(ignore_index=255), use_graph=use_graph)
    assert tr.fuse_head_loss
    x, y = synthetic_batch(2, 64, 128)
    x, y = x.to(DEV), y.to(DEV)
    losses = [tr.step_async(x, y).item() for _ in range(12)]
    assert all(np.isfinite(losses)), losses
    assert min(losses[-3:]) < losses[0], losses
    bn = m.decoder.level4[1]
    assert int(bn.num_batches_tracked) == 12 and bn.running_var.ne(1).any()


def test_esnet_whole_model_eval_logits_and_argmax(golden_dir):
    """ESNet(3, 19) (TSS/models/esnet.py:8-44) on the 2 x 3 x 32 x 64 lattice image: state_dict keys and shapes of the reference,
    eval-mode logits and arg-max against the imported reference's fixture, train-mode logits, finite gradients everywhere."""
    import torch_semantic_segmentation_amd as tssa
    g = cases.load_npz(os.path.join(golden_dir, 'zoo.npz'))
    m = cases.product_zoo('es_net')
    o = cases.oracle_zoo('es_net')
    assert list(m.state_dict()) == list(o.state_dict())
    assert all(tuple(a.shape) == tuple(b.shape) for a, b in zip(m.state_dict().values(), o.state_dict().values()))
    m.load_state_dict(formula_state(m), strict=True)
    cases.zero_all_dropout(m)
    m.to(DEV).eval()
    tssa.set_compute_dtype(m, torch.float32)
    x = lattice_input(*cases.ESNET_SHAPE).to(DEV)
    with torch.no_grad():
        out = m(x)
    assert tuple(out.shape) == (2, 19, 32, 64)
    assert close(out[:, :, ::2, ::2].float().cpu().numpy(), g['eval/es_net/out_sub2'])
    mism = (out.argmax(1).cpu().numpy().astype(np.uint8) != g['eval/es_net/argmax']).mean()
    assert mism <= 2e-3, mism
    m.train()
    out_t = m(x)
    # ill-conditioned in train mode on these tiny maps: anchored on the reference in f64, bounded by 3x the reference's own f32 distance
    bound = max(5e-3, 3 * float(g['train/es_net/err_ref32']))
    assert close(out_t[:, :, ::2, ::2].detach().float().cpu().numpy(), g['train/es_net/out64_sub2'], rel=bound), bound
    out_t.float().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


@pytest.mark.parametrize('B', [1, 2])
def test_aspp_project_layer_reads_its_branches_in_place(B):
    """Eval-mode ASPP in bf16: the 1x1 project layer walks the five branches source by source (tss_pwconv_fwd_multi: every branch's
    BatchNorm + ReLU on load, the image-pooling row broadcast, no concat buffer) -- same output as the concat path that the same
    module takes with gradients enabled; with two images the pooled branch is not one row and the module must fall back by itself."""
    import torch_semantic_segmentation_amd as tssa
    from torch_semantic_segmentation_amd import ops
    from torch_semantic_segmentation_amd.models.aspp import ASPP
    torch.manual_seed(1)
    m = ASPP(128, 128, atrous_rates=(2, 4, 6)).to(DEV)
    for mod in m.modules():
        if isinstance(mod, nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.3); mod.running_var.uniform_(0.5, 1.5); mod.weight.data.uniform_(0.5, 1.5); mod.bias.data.normal_(0, 0.2)
    tssa.set_compute_dtype(m, torch.bfloat16)
    m.eval()
    x = ops.to_nhwc(torch.randn(B, 128, 40, 72, device=DEV).to(torch.bfloat16))
    calls = []
    real = ops.conv_unit_multi
    ops.conv_unit_multi = lambda *a, **k: calls.append(real(*a, **k)) or calls[-1]
    try:
        with torch.no_grad():
            y1 = m(x).float()
    finally:
        ops.conv_unit_multi = real
    assert len(calls) == 1 and (calls[0] is not None) == (B == 1)
    y0 = m(x).detach().float()                      # gradients enabled: the concat path
    assert torch.isfinite(y1).all()
    assert rel(y1, y0) < 4e-3, rel(y1, y0)



@pytest.mark.parametrize('name', ['led_net', 'es_net'])
def test_zoo_whole_model_frozen_gradients_vs_reference(golden_dir, name):
    """Whole LedNet / ESNet, one backward pass with the BatchNorms frozen, f32 (round 4, VERDICT r03 weak 3: the
    whole models had eval logits and 'every gradient finite' only).  Fixture tests/golden/zoo_frozen.npz: the imported reference with
    default init (seed 0) on the seeded N(0,1) batch, in f32 AND in f64 -- these 40-layer ReLU / max-pool stacks are ill-conditioned at
    any size we can afford (the reference's own f32 gradients sit 0.5 - 5 % from its f64 ones, tensor by tensor), so, as for G3c, the
    f64 run is the anchor: EVERY parameter's gradient within 3 x the reference's own f32 distance of the f64 gradient (norms for all
    tensors, full tensors for five), the loss within 3 x its f32 distance.  The weights come from the oracle built under the same seed
    (bit-identical to the reference's: tests/test_oracle_golden.py)."""
    import torch_semantic_segmentation_amd as tssa
    g = cases.load_npz(os.path.join(golden_dir, 'zoo_frozen.npz'))
    torch.manual_seed(0)
    o = cases.oracle_zoo(name)
    m = cases.product_zoo(name)
    m.load_state_dict(o.state_dict(), strict=True)
    cases.zero_all_dropout(m)
    cases.load_fixture_buffers(m, g, name)
    m.to(DEV).eval()
    tssa.set_compute_dtype(m, torch.float32)
    x, y = synthetic_batch(2, 64, 128)
    loss = tssa.CrossEntropyLoss(ignore_index=255)(m(x.to(DEV)), y.to(DEV))
    loss.backward()
    l32, l64 = float(g[name + '/loss32']), float(g[name + '/loss64'])
    assert abs(loss.item() - l64) <= 3 * abs(l32 - l64) + 1e-5 * abs(l64), (loss.item(), l32, l64)
    names = [n for n, _ in m.named_parameters()]
    norms = np.array([p.grad.double().norm().item() for _, p in m.named_parameters()])
    n64, e32 = g[name + '/grad_norms64'], g[name + '/err_ref32_per_tensor']
    assert len(norms) == len(n64)
    # a gradient norm can differ from the f64 one by at most the distance of the two gradients
    bound = 3 * e32 * n64 + 1e-3 * n64.max()
    worst = int((np.abs(norms - n64) / bound).argmax())
    assert (np.abs(norms - n64) <= bound).all(), (names[worst], norms[worst], n64[worst], e32[worst])
    checked = 0
    for key in g:
        if key.startswith(name + '/grad64.'):
            pname = key[len(name) + 8:]
            i = names.index(pname)
            got = m.get_parameter(pname).grad.double().cpu().numpy()
            err = np.linalg.norm(got - g[key]) / max(np.linalg.norm(g[key]), 1e-30)
            # floor: ReLU decisions at the logits.  ESNet's classifier ends in BatchNorm + ReLU, and an f32 forward opens / closes a handful of
            # the 311 296 output units differently from the f64 run (measured: torch f32 5, HIP f32 6 -- tools/zoo_fwd_diag.py); a unit at
            # a LABEL pixel carries (p - 1) / N ~ 1 / 15 579 of the cross-entropy gradient, 6e-4 of |d(bn.bias)| ~ 0.1, so which units flip
            # (luck, not accuracy: same logits error 7e-5 vs 6e-5, same d(logits) error 1.9e-5 vs 1.6e-5, every kernel of the block at
            # 1e-7 of f64 on the same input -- tools/cls_diag.py, tools/zoo_mix2_diag.py) moves the last layers' gradients by 1e-3 - 3e-3.
            # The reference's own f32 distance does not cover this for the last block, hence a floor of a few flips' worth.
            assert err <= 3 * e32[i] + 6e-3, (pname, err, e32[i])
            checked += 1
    assert checked == 5
    print(name, 'HIP f32 vs reference f64: loss', loss.item(), l64, ' max |norm - norm64| / bound', float((np.abs(norms - n64) / bound).max()))
