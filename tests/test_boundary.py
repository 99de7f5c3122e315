"""CPU (no GPU): the drop-in boundary -- C-ABI symbols, module tree / state_dict contract, host-side logic."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
from torch import nn
from torch.nn.modules.batchnorm import _BatchNorm

from oracle import nets as O
from oracle.recipe import formula_state
from tests import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from torch_semantic_segmentation_amd import _native as N
    header = open(os.path.join(ROOT, 'include', 'tss_hip.h')).read()
    header = re.sub(r'/\*.*?\*/', ' ', header, flags=re.S)
    declared = set(re.findall(r'\b(tss_\w+)\s*\(', header))
    assert len(declared) >= 40
    if not os.path.exists(N.LIB_PATH):
        pytest.skip('libtss_hip.so not built here (driver runs build() first)')
    handle = ctypes.CDLL(N.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(handle, s)]
    assert not missing, missing
    lib = N.lib()                              # binds argtypes for every declared function
    assert lib.tss_arch() == b'gfx950' and lib.tss_version() == 1
    assert set(N.parse_header()) <= declared
    assert {'tss_pwconv_bwd_weight_ws', 'tss_upsample_ce_fwd', 'tss_upsample_ce_bwd', 'tss_prof_records'} <= declared
    assert lib.tss_pwconv_bwd_weight_ws(4096, 64, 128, 1) > 0 and lib.tss_pwconv_bwd_weight_ws(4096, 64, 128, 0) == 0


def test_missing_library_fails_loudly(monkeypatch):
    from torch_semantic_segmentation_amd import _native as N
    monkeypatch.setattr(N, '_lib', None)
    monkeypatch.setattr(N, 'LIB_PATH', '/nonexistent/libtss_hip.so')
    with pytest.raises(RuntimeError, match='no non-HIP fallback'):
        N.lib()


@pytest.mark.parametrize('name', cases.MODEL_NAMES)
def test_state_dict_is_interchangeable_with_the_reference_layout(name):
    m, o = cases.product_model(name), O.build(name)
    a, b = m.state_dict(), o.state_dict()
    assert list(a) == list(b)
    assert all(a[k].shape == b[k].shape and a[k].dtype == b[k].dtype for k in a)
    m.load_state_dict(formula_state(o), strict=True)          # strict load, as scripts/train_fastscnn.py:119-121
    o.load_state_dict(m.state_dict(), strict=True)
    assert [n for n, _ in m.named_parameters()] == [n for n, _ in o.named_parameters()]
    assert all(isinstance(x, _BatchNorm) for x in m.modules() if type(x).__name__.startswith('BatchNorm'))
    assert sum(isinstance(x, _BatchNorm) for x in m.modules()) == sum(isinstance(x, _BatchNorm) for x in o.modules())


def test_public_surface_matches_reference_names():
    import importlib
    F = importlib.import_module('torch_semantic_segmentation_amd.models.fastscnn')
    C = importlib.import_module('torch_semantic_segmentation_amd.models.contextnet')
    for n in ('FastSCNN', 'fastscnn', 'Classifier', 'Conv2dBlock', 'DWConv2dBlock', 'DSConv2dBlock', 'BottleneckBlock',
              'BottleneckModule', 'PyramidPoolingModule', 'FeatureFusionModule'):
        assert hasattr(F, n), n
    for n in ('ContextNet', 'contextnet12', 'contextnet14', 'contextnet18', 'Classifier', 'LinearBottleneck',
              'FeatureFusionModule', 'BottleneckBlock', 'DWConvBlock', 'ConvBlock'):
        assert hasattr(C, n), n
    m = F.fastscnn(3, 19)
    for attr in ('downsample', 'features', 'fusion', 'classifier'):
        assert isinstance(getattr(m, attr), nn.Module)
    c = C.contextnet14(3, 19)
    assert c.scale_factor == 4 and C.contextnet12(3, 19).scale_factor == 2 and C.contextnet18(3, 19).scale_factor == 8
    for attr in ('spatial', 'context', 'feature_fusion', 'classifier'):
        assert isinstance(getattr(c, attr), nn.Module)
    with pytest.raises(ValueError, match='must be the same in depthwise'):
        C.DWConvBlock(32, 48, 3)                               # TSS/models/contextnet.py:153-155


def test_no_cpu_fallback():
    m = cases.product_model('contextnet14')
    with pytest.raises(RuntimeError, match='HIP path only'):
        m(torch.zeros(1, 3, 64, 64))
    from torch_semantic_segmentation_amd import ops
    with pytest.raises(RuntimeError, match='HIP path only'):
        ops.cross_entropy(torch.zeros(1, 19, 8, 8), torch.zeros(1, 8, 8, dtype=torch.int64))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'torch_semantic_segmentation_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f


def test_compute_dtype_switch_and_layout_helpers():
    import torch_semantic_segmentation_amd as tssa
    from torch_semantic_segmentation_amd import ops
    m = cases.product_model('fastscnn')
    tssa.set_compute_dtype(m, torch.bfloat16)
    assert m.downsample[0].act_dtype == torch.bfloat16 and all(p.dtype == torch.float32 for p in m.parameters())
    with pytest.raises(TypeError):
        tssa.set_compute_dtype(m, torch.float16)
    assert ops.round_up(19, 8) == 24 and ops._out_size(torch.zeros(1, 1, 128, 256), None, 1 / 4) == (32, 64)
    assert ops._out_size(torch.zeros(1, 1, 6, 10), (12, 7), None) == (12, 7)


def test_flat_adamw_aliases_parameters_and_gradients():
    from torch_semantic_segmentation_amd import engine as E
    m = cases.product_model('fastscnn')
    before = {k: v.clone() for k, v in m.state_dict().items()}
    opt = E.FlatAdamW(m.parameters(), lr=1e-3, weight_decay=1e-5)
    assert opt.flat_param.numel() == 1137795 == opt.flat_grad.numel()
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    p = m.classifier[3].bias
    opt.flat_grad.fill_(2.0)
    assert torch.equal(p.grad, torch.full_like(p, 2.0))
    opt.flat_param.zero_()
    assert float(p.abs().sum()) == 0.0
    opt.zero_grad()
    assert float(opt.flat_grad.abs().sum()) == 0.0
    with pytest.raises(RuntimeError, match='HIP path only'):
        opt.step()


def test_metrics_and_sharding_helpers():
    from torch_semantic_segmentation_amd import engine as E
    cm = torch.tensor([[5., 1.], [2., 8.]], dtype=torch.float64)
    met = E.confusion_metrics(cm)
    assert np.isclose(met['accuracy'], 13 / 16)
    assert np.allclose(met['iou'].numpy(), [5 / 8, 8 / 11])
    assert np.isclose(met['miou'], (5 / 8 + 8 / 11) / 2)
    assert np.allclose(met['dice'].numpy(), [10 / 13, 16 / 19])
    assert E.shard_batch(10, 4, 1) == [1, 5, 9]
    assert sorted(sum((E.shard_batch(64, 8, r) for r in range(8)), [])) == list(range(64))
    assert E.setup_distributed(enable=False) == (1, 0, 0)


def test_deep_supervision_wrapper_contract():
    from torch_semantic_segmentation_amd import engine as E

    class Toy(nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b = nn.Conv2d(3, 4, 1), nn.Conv2d(4, 2, 1)

        def forward(self, x):
            return self.b(self.a(x))
    toy = Toy()
    w = E.DeepSupervisionWrapper(toy, [(toy.a, nn.Conv2d(4, 5, 1))])
    assert [k.split('.')[0] for k in w.state_dict()][:1] == ['module'] and any(k.startswith('auxiliary.0.') for k in w.state_dict())
    x = torch.randn(2, 3, 4, 4)
    out, aux = w.train()(x)
    assert out.shape == (2, 2, 4, 4) and aux[0].shape == (2, 5, 4, 4) and not toy.a._forward_hooks
    assert w.eval()(x).shape == (2, 2, 4, 4)


def test_convert_syncbn_model_keeps_modules_parameters_and_state_dict():
    """apex.parallel.convert_syncbn_model stand-in (scripts/train_fastscnn.py:144-145): same module objects, same
    parameters/buffers, same state_dict keys; the layers are still _BatchNorm instances; without an initialised process
    group (or in eval mode) they behave as ordinary BatchNorm."""
    import importlib
    import torch
    import torch_semantic_segmentation_amd as tssa
    from torch_semantic_segmentation_amd import ops
    F = importlib.import_module('torch_semantic_segmentation_amd.models.fastscnn')
    model = F.FastSCNN(3, 19)
    keys, params = list(model.state_dict().keys()), [id(p) for p in model.parameters()]
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    assert tssa.convert_syncbn_model(model) is model
    assert list(model.state_dict().keys()) == keys and [id(p) for p in model.parameters()] == params
    assert len(bns) == 44 and all(isinstance(m, tssa.SyncBatchNorm) and isinstance(m, torch.nn.modules.batchnorm._BatchNorm) for m in bns)
    assert all(ops._sync_group(m) is None for m in bns)        # no process group here: local statistics


def test_half_and_to_dtype_select_bf16_activations_and_back():
    """`.half()` / `.to(dtype)` on the modules (TSS scripts/contextnet/benchmark_contextnet.py:62).
    Host-side part: the cast is accepted, parameters really are 16-bit, the activation format of the HIP path follows."""
    from torch_semantic_segmentation_amd.models._fused import FusedSequential
    m = cases.product_model('contextnet14')
    assert getattr(m, 'compute_dtype', None) is None
    m.half()
    assert all(p.dtype == torch.float16 for p in m.parameters())
    assert m.compute_dtype == torch.bfloat16 and all(f.act_dtype == torch.bfloat16 for f in m.modules() if isinstance(f, FusedSequential))
    m.float()
    assert m.compute_dtype == torch.float32
    m.to(torch.bfloat16)
    assert m.compute_dtype == torch.bfloat16 and next(m.parameters()).dtype == torch.bfloat16
    assert len(m.state_dict()) == 314                      # same keys whatever the dtype


def test_flat_adamw_state_dict_round_trip_and_detached_gradients_raise():
    """the flat moments / step counter travel through state_dict, and step() refuses to run on gradients that
    no longer alias the flat buffer (model.zero_grad(set_to_none=True) would otherwise make it apply zeros silently)."""
    from torch_semantic_segmentation_amd import engine as E
    m = nn.Sequential(nn.Conv2d(3, 4, 1), nn.BatchNorm2d(4))
    opt = E.FlatAdamW(m.parameters(), lr=1e-3)
    opt.exp_avg.fill_(0.5); opt.exp_avg_sq.fill_(0.25); opt.step_count = 3
    sd = opt.state_dict()
    assert float(sd['flat']['state_vec'][0]) == 3.0 and abs(float(sd['flat']['state_vec'][1]) - (1 - 0.9 ** 3)) < 1e-6
    opt2 = E.FlatAdamW(nn.Sequential(nn.Conv2d(3, 4, 1), nn.BatchNorm2d(4)).parameters(), lr=1e-3)
    opt2.load_state_dict(sd)
    assert torch.equal(opt2.exp_avg, opt.exp_avg) and torch.equal(opt2.exp_avg_sq, opt.exp_avg_sq)
    assert opt2.step_count == 3 and torch.equal(opt2.state_vec, sd['flat']['state_vec'])
    # a torch.optim.AdamW checkpoint (per-parameter state, no flat moments) must not load silently
    ref = torch.optim.AdamW(nn.Sequential(nn.Conv2d(3, 4, 1), nn.BatchNorm2d(4)).parameters(), lr=1e-3)
    for p_ in ref.param_groups[0]['params']:
        p_.grad = torch.zeros_like(p_)
    ref.step()
    with pytest.raises(ValueError, match='no flat moments'):
        opt2.load_state_dict(ref.state_dict())
    opt._check_aliases()
    m.zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError, match='no longer aliases'):
        opt._check_aliases()
    opt.reattach()
    opt._check_aliases()


def test_no_compiler_copy_reads_a_pending_row_register():
    """csrc/dwroll.hip requests its rows with inline assembly and hand-placed waits; a compiler-made copy of a request's
    destination register in front of the wait would read a row that has not arrived (tools/check_pending_regs.py scans the
    gfx950 assembly for exactly that: it happened once, and only showed as statistics that were off by 1e-3 on some runs)."""
    import importlib.util
    import os
    import shutil
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        import pytest
        pytest.skip('hipcc not available')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('check_pending_regs', os.path.join(root, 'tools', 'check_pending_regs.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main() == 0
