"""GPU parity of the SURVEY.md section 8f N4 rows: the PSPNet head (TSS/models/pspnet.py) against golden vectors generated from
the reference itself, and the DeepLab-style ASPP head of BASELINE config 5 against plain torch (the reference has no ASPP:
its parity is pinned to torch.nn.functional, oracle/aspp.py says so)."""
import os

import numpy as np
import pytest
import torch
from torch import nn

from oracle import aspp as OA
from oracle.recipe import formula_state, lattice_input, synthetic_batch
from tests import cases

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


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
