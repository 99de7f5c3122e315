"""CPU: the oracle restatement reproduces the reference-generated golden vectors (bit for bit)."""
import os

import numpy as np
import pytest
import torch
from torch import nn

from oracle import nets as O
from oracle.recipe import formula_state, lattice_input, lattice_target, train_step
from tests import cases

torch.set_num_threads(4)


@pytest.fixture(scope='module')
def blocks(golden_dir):
    return {m: cases.load_npz(os.path.join(golden_dir, 'blocks_%s.npz' % m)) for m in ('train', 'eval')}


@pytest.mark.parametrize('mode', ['train', 'eval'])
@pytest.mark.parametrize('name', sorted(cases.BLOCK_SHAPES))
def test_block_bit_identical(blocks, name, mode):
    g = blocks[mode]
    m = cases.oracle_block(name)
    m.load_state_dict(formula_state(m), strict=True)
    cases.zero_dropout(m)
    m.train(mode == 'train')
    xs = [x.requires_grad_(True) for x in cases.block_inputs(name)]
    out = m(*xs)
    out.backward(cases.block_cotangent(out.shape))
    assert np.array_equal(out.detach().numpy(), g[name + '/out'])
    for i, x in enumerate(xs):
        assert np.array_equal(x.grad.numpy(), g['%s/dx%d' % (name, i)])
    for pname, p in m.named_parameters():
        assert np.array_equal(p.grad.numpy(), g['%s/dw.%s' % (name, pname)]), pname
    if mode == 'train':
        for bname, b in m.named_buffers():
            if bname.endswith(('running_mean', 'running_var')):
                assert np.array_equal(b.numpy(), g['%s/buf.%s' % (name, bname)]), bname


@pytest.mark.parametrize('name', cases.MODEL_NAMES)
def test_model_eval_bit_identical(golden_dir, name):
    g = cases.load_npz(os.path.join(golden_dir, 'eval_models.npz'))
    m = O.build(name)
    m.load_state_dict(formula_state(m, gain=1.0), strict=True)
    m.eval()
    with torch.no_grad():
        logits = m(lattice_input(*cases.EVAL_SHAPE))
    assert np.array_equal(logits[:, :, ::4, ::4].numpy(), g[name + '/sub'])
    assert np.array_equal(logits.argmax(1).to(torch.uint8).numpy(), g[name + '/argmax'])
    s = np.array([logits.sum().item(), logits.abs().sum().item()])
    assert np.array_equal(s, g[name + '/sum_abs'])


@pytest.mark.parametrize('name', ['fastscnn', 'contextnet14'])
def test_train_steps_bit_identical(golden_dir, name):
    g = cases.load_npz(os.path.join(golden_dir, 'train_steps.npz'))
    m = O.build(name)
    m.load_state_dict(formula_state(m), strict=True)
    cases.zero_dropout(m)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-5)
    loss_fn = nn.CrossEntropyLoss(ignore_index=255)
    x = lattice_input(*cases.TRAIN_SHAPE)
    y = lattice_target(cases.TRAIN_SHAPE[0], cases.TRAIN_SHAPE[2], cases.TRAIN_SHAPE[3])
    losses = [train_step(m, opt, loss_fn, x, y)]
    norms = np.array([p.grad.double().norm().item() for p in m.parameters()])
    assert np.array_equal(norms, g[name + '/grad_norms'])
    for key in g:
        if key.startswith(name + '/grad.'):
            pname = key[len(name) + 6:]
            assert np.array_equal(m.get_parameter(pname).grad.numpy(), g[key]), pname
    losses.append(train_step(m, opt, loss_fn, x, y))
    assert np.array_equal(np.array(losses), g[name + '/losses'])
    after2 = np.array([p.detach().double().norm().item() for p in m.parameters()])
    assert np.array_equal(after2, g[name + '/param_norms_after2'])


def frozen_step(m, opt, loss_fn, x, y):
    m.eval()
    opt.zero_grad()
    loss = loss_fn(m(x), y)
    loss.backward()
    opt.step()
    return loss.item()


@pytest.mark.parametrize('name', cases.MODEL_NAMES)
def test_frozen_bn_steps_bit_identical(golden_dir, name):
    g = cases.load_npz(os.path.join(golden_dir, 'frozen_steps.npz'))
    m = O.build(name)
    m.load_state_dict(formula_state(m), strict=True)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-5)
    loss_fn = nn.CrossEntropyLoss(ignore_index=255)
    x = lattice_input(*cases.TRAIN_SHAPE)
    y = lattice_target(cases.TRAIN_SHAPE[0], cases.TRAIN_SHAPE[2], cases.TRAIN_SHAPE[3])
    losses = [frozen_step(m, opt, loss_fn, x, y)]
    norms = np.array([p.grad.double().norm().item() for p in m.parameters()])
    assert np.array_equal(norms, g[name + '/grad_norms'])
    losses.append(frozen_step(m, opt, loss_fn, x, y))
    assert np.array_equal(np.array(losses), g[name + '/losses'])


def test_state_dict_contract():
    """Key/shape contract quoted in SURVEY.md §5 (266 / 314 keys, parameter counts)."""
    f = O.build('fastscnn')
    c = O.build('contextnet14')
    assert len(f.state_dict()) == 266 and sum(p.numel() for p in f.parameters()) == 1137795
    assert len(c.state_dict()) == 314 and sum(p.numel() for p in c.parameters()) == 1024019
    assert f.state_dict()['downsample.0.0.weight'].shape == (32, 3, 3, 3)
    assert f.state_dict()['classifier.3.bias'].shape == (19,)


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='reference only exists in the build container')
def test_oracle_keys_equal_reference():
    import importlib
    import sys
    sys.path.insert(0, '/root/reference')
    try:
        rf = importlib.import_module('torch_semantic_segmentation.models.fastscnn')
        rc = importlib.import_module('torch_semantic_segmentation.models.contextnet')
    finally:
        sys.path.remove('/root/reference')
    for ours, theirs in ((O.build('fastscnn'), rf.fastscnn(3, 19)), (O.build('contextnet12'), rc.contextnet12(3, 19))):
        a, b = ours.state_dict(), theirs.state_dict()
        assert list(a) == list(b)
        assert all(a[k].shape == b[k].shape and a[k].dtype == b[k].dtype for k in a)


@pytest.mark.parametrize('name', ['fastscnn', 'contextnet14'])
def test_seeded_default_init_step_bit_identical(golden_dir, name):
    """G3c: default init under torch.manual_seed(0), seeded N(0,1) batch at 2x3x96x160, one train-mode forward/backward in
    f32 and in f64 -- the oracle reproduces the reference's losses, per-parameter gradient norms and full gradients."""
    from oracle.recipe import synthetic_batch
    g = cases.load_npz(os.path.join(golden_dir, 'train_seeded.npz'))
    x, y = synthetic_batch(2, 96, 160)
    for dt, tag in ((torch.float32, '32'), (torch.float64, '64')):
        torch.manual_seed(0)
        m = O.build(name)
        cases.zero_dropout(m)
        m.to(dt).train()
        loss = nn.CrossEntropyLoss(ignore_index=255)(m(x.to(dt)), y)
        loss.backward()
        assert loss.item() == float(g['%s/loss%s' % (name, tag)])
        norms = np.array([p.grad.double().norm().item() for p in m.parameters()])
        assert np.array_equal(norms, g['%s/grad_norms%s' % (name, tag)])
        if tag == '64':
            for key in g:
                if key.startswith(name + '/grad64.'):
                    assert np.array_equal(m.get_parameter(key[len(name) + 8:]).grad.numpy(), g[key]), key


@pytest.mark.parametrize('mode', ['train', 'eval'])
@pytest.mark.parametrize('name', sorted(cases.PSP_SHAPES))
def test_pspnet_head_bit_identical(golden_dir, name, mode):
    """oracle/aspp.py PyramidPools / PSPNetOracle reproduce TSS/models/pspnet.py (imported by make_golden.py gen_pspnet)."""
    g = cases.load_npz(os.path.join(golden_dir, 'pspnet.npz'))
    m = cases.oracle_psp(name)
    m.load_state_dict(formula_state(m), strict=True)
    m.train(mode == 'train')
    xs = [x.requires_grad_(True) for x in cases.psp_inputs(name)]
    out = m(*xs)
    out.backward(cases.block_cotangent(out.shape))
    key = '%s/%s/' % (mode, name)
    assert np.array_equal(out.detach().numpy(), g[key + 'out'])
    assert np.array_equal(xs[0].grad.numpy(), g[key + 'dx0'])
    for pname, p in m.named_parameters():
        assert np.array_equal(p.grad.numpy(), g[key + 'dw.' + pname]), pname


@pytest.mark.parametrize('mode', ['train', 'eval'])
@pytest.mark.parametrize('name', sorted(cases.ZOO_SHAPES))
def test_lednet_esnet_blocks_bit_identical(golden_dir, name, mode):
    """oracle/zoo.py restates TSS/models/lednet.py (DownsamplingBlock, APNModule) and TSS/models/esnet.py (FCUBlock, FPCUBlock,
    DownsamplingBlock): same bits as the imported reference (make_golden.py gen_zoo) in forward, dX and every parameter gradient."""
    g = cases.load_npz(os.path.join(golden_dir, 'zoo.npz'))
    m = cases.oracle_zoo(name)
    m.load_state_dict(formula_state(m), strict=True)
    cases.zero_all_dropout(m)
    m.train(mode == 'train')
    xs = [x.requires_grad_(True) for x in cases.zoo_inputs(name)]
    out = m(*xs)
    out.backward(cases.block_cotangent(out.shape))
    key = '%s/%s/' % (mode, name)
    assert np.array_equal(out.detach().numpy(), g[key + 'out'])
    assert np.array_equal(xs[0].grad.numpy(), g[key + 'dx0'])
    for pname, p in m.named_parameters():
        assert np.array_equal(p.grad.numpy(), g[key + 'dw.' + pname]), pname


def test_lednet_whole_model_bit_identical(golden_dir):
    """oracle/zoo.py LedNetOracle == TSS/models/lednet.py LedNet: eval logits, arg-max and train-mode logits of the golden fixture."""
    from oracle.recipe import lattice_input
    g = cases.load_npz(os.path.join(golden_dir, 'zoo.npz'))
    m = cases.oracle_zoo('led_net')
    m.load_state_dict(formula_state(m), strict=True)
    cases.zero_all_dropout(m)
    x = lattice_input(*cases.LEDNET_SHAPE)
    m.eval()
    with torch.no_grad():
        out = m(x)
    assert np.array_equal(out[:, :, ::4, ::4].numpy(), g['eval/led_net/out_sub4'])
    assert np.array_equal(out.argmax(1).numpy().astype(np.uint8), g['eval/led_net/argmax'])
    m.train()
    out = m(x)
    assert np.array_equal(out[:, :, ::4, ::4].detach().numpy(), g['train/led_net/out_sub4'])


def test_esnet_whole_model_bit_identical(golden_dir):
    """oracle/zoo.py ESNetOracle == TSS/models/esnet.py ESNet on the golden fixture (eval logits, arg-max, train-mode logits)."""
    from oracle.recipe import lattice_input
    g = cases.load_npz(os.path.join(golden_dir, 'zoo.npz'))
    m = cases.oracle_zoo('es_net')
    m.load_state_dict(formula_state(m), strict=True)
    cases.zero_all_dropout(m)
    x = lattice_input(*cases.ESNET_SHAPE)
    m.eval()
    with torch.no_grad():
        out = m(x)
    assert np.array_equal(out[:, :, ::2, ::2].numpy(), g['eval/es_net/out_sub2'])
    assert np.array_equal(out.argmax(1).numpy().astype(np.uint8), g['eval/es_net/argmax'])
    m.train()
    out = m(x)
    assert np.array_equal(out[:, :, ::2, ::2].detach().numpy(), g['train/es_net/out_sub2'])



@pytest.mark.parametrize('name', ['led_net', 'es_net'])
def test_zoo_whole_model_frozen_gradients_bit_identical(golden_dir, name):
    """oracle/zoo.py's whole LedNet / ESNet == the imported reference (tests/golden/make_golden.py gen_zoo_frozen, round 4): built under
    torch.manual_seed(0) it has the reference's default-init weights (per-tensor checksums of the fixture), and one backward pass with
    the BatchNorms frozen on the fixture's statistics gives the reference's f32 loss and EVERY parameter's gradient norm, bit for bit."""
    from oracle.recipe import synthetic_batch
    g = cases.load_npz(os.path.join(golden_dir, 'zoo_frozen.npz'))
    torch.manual_seed(0)
    m = cases.oracle_zoo(name)
    cases.zero_all_dropout(m)
    assert np.array_equal(np.array([p.detach().double().abs().sum().item() for p in m.parameters()]), g[name + '/wsum'])
    cases.load_fixture_buffers(m, g, name)
    m.eval()
    x, y = synthetic_batch(2, 64, 128)
    loss = torch.nn.CrossEntropyLoss(ignore_index=255)(m(x), y)
    loss.backward()
    assert loss.item() == float(g[name + '/loss32'])
    norms = np.array([p.grad.double().norm().item() for _, p in m.named_parameters()])
    assert np.array_equal(norms, g[name + '/grad_norms32'])
