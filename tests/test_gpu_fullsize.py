"""GPU parity at the sizes BASELINE.json names ("no -m gpu test touches a BASELINE-sized tensor").

  config 1  4 x 3 x 512 x 1024 f32, seeded N(0,1) inputs, default init: FastSCNN and ContextNet14 against the CPU oracle --
            the north_star's literal sentence ("logits within 1e-3 rel fp32, argmax masks bit-exact"), eval AND train
            mode, plus the train-mode loss and gradients of FastSCNN (f64 oracle, conditioning-aware bound).
  config 2/3  8 x 3 x 1024 x 2048 bf16 train step through Trainer(use_graph=True) (the benchmarked path: lean kernels,
            fused head + loss, FlatAdamW, HIP-graph replay) against the f32 general-kernel path of the same library at
            the same size, with the general bf16 kernels as the yardstick.
  config 5  1 x 3 x 2048 x 4096 eval: f32 logits against the CPU oracle, fused argmax + confusion == unfused, bf16 lean
            forward against f32 within the bf16 yardstick.

The CPU oracle needs ~1-4 s per forward at these sizes (16 host threads); every case runs it once.
"""
import copy

import numpy as np
import pytest
import torch
from torch import nn

from oracle import nets as O
from oracle.recipe import synthetic_batch
from tests import cases

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _pair(name, seed=0):
    torch.manual_seed(seed)
    ref = O.build(name)
    hip = cases.product_model(name)
    hip.load_state_dict(ref.state_dict(), strict=True)
    cases.zero_dropout(ref)
    cases.zero_dropout(hip)
    return ref, hip.to(DEV)


def _argmax_check(logits_hip, logits_ref, pred_hip):
    """argmax masks bit-exact, except where the ORACLE's own top-2 gap is below what 1e-3-relative logits can resolve."""
    ref_arg = logits_ref.argmax(1)
    top2 = logits_ref.topk(2, dim=1).values
    gap = (top2[:, 0] - top2[:, 1])
    mism = pred_hip.cpu().long() != ref_arg
    err = (logits_hip.float().cpu() - logits_ref).abs().max().item()
    assert (gap[mism] <= 2 * err + 1e-7).all(), 'argmax differs at a pixel whose top-2 gap exceeds twice the logit error'
    return mism.float().mean().item()


@pytest.mark.parametrize('name', ['fastscnn', 'contextnet14'])
def test_config1_eval_logits_and_argmax_vs_cpu_oracle(name):
    import torch_semantic_segmentation_amd as tssa
    ref, hip = _pair(name)
    x, _ = synthetic_batch(4, 512, 1024)
    ref.eval(); hip.eval()
    with torch.no_grad():
        want = ref(x)
        got = hip(x.to(DEV))
        pred, _ = tssa.argmax_confusion(got)
    assert tuple(got.shape) == (4, 19, 512, 1024) and got.dtype == torch.float32
    rel = cases.rel_err(got.cpu().numpy(), want.numpy())
    assert rel < 1e-3, rel
    frac = _argmax_check(got, want, pred)
    assert frac < 1e-4, frac
    print('%s config-1 eval: logits rel err %.2e, argmax mismatch fraction %.2e' % (name, rel, frac))


@pytest.mark.parametrize('name', ['fastscnn', 'contextnet14'])
def test_config1_train_mode_forward_vs_cpu_oracle(name):
    """Batch-statistics BatchNorm at config 1: logits within 1e-3 of the f32 oracle (FastSCNN) / within 3x the oracle's own
    f32-vs-f64 distance (ContextNet: its 40-layer context branch amplifies rounding, tests/golden/make_golden.py G3c),
    loss to 1e-4, running statistics after the step."""
    import torch_semantic_segmentation_amd as tssa
    ref, hip = _pair(name)
    x, y = synthetic_batch(4, 512, 1024)
    ref.train(); hip.train()
    with torch.no_grad():
        want = ref(x)
        # the f32 run above already updated the running statistics of `ref`: a fresh copy (same seed) for the f64 run
        torch.manual_seed(0)
        fresh = O.build(name)
        cases.zero_dropout(fresh)
        fresh.double().train()
        want64 = fresh(x.double())
        got = hip(x.to(DEV))
    e_ref32 = cases.rel_err(want.numpy(), want64.numpy())
    e_hip = cases.rel_err(got.cpu().numpy(), want64.numpy())
    print('%s config-1 train forward: |hip - f64| %.2e, |oracle f32 - f64| %.2e' % (name, e_hip, e_ref32))
    assert e_hip <= max(1e-3, 3 * e_ref32), (e_hip, e_ref32)
    loss_r = nn.functional.cross_entropy(want64, y, ignore_index=255).item()
    loss_h = tssa.cross_entropy(got, y.to(DEV), ignore_index=255).item()
    assert abs(loss_h / loss_r - 1) < 1e-4
    for (k, b), (_, br) in zip(hip.named_buffers(), fresh.named_buffers()):
        if k.endswith(('running_mean', 'running_var')):
            # (a BatchNorm that follows a zero-mean input through a linear conv has an analytically zero mean: absolute floor)
            d = np.abs(b.double().cpu().numpy() - br.numpy()).max()
            assert d <= max(1e-3, 3 * e_ref32) * np.abs(br.numpy()).max() + 1e-6, k
        elif k.endswith('num_batches_tracked'):
            assert int(b) == 1


def test_config1_fastscnn_train_step_gradients_vs_f64_oracle():
    """BASELINE config 1 as a whole: forward + CE(ignore 255) + backward, f32, against the f64 oracle; the bound is 3x the
    f32 oracle's own distance from f64 at this size (measured here, printed), floor 2e-3."""
    import torch_semantic_segmentation_amd as tssa
    x, y = synthetic_batch(4, 512, 1024)
    loss_fn = nn.CrossEntropyLoss(ignore_index=255)
    grads = {}
    for dt in (torch.float32, torch.float64):
        torch.manual_seed(0)
        m = O.build('fastscnn')
        cases.zero_dropout(m)
        m.to(dt).train()
        loss = loss_fn(m(x.to(dt)), y)
        loss.backward()
        grads[dt] = (torch.cat([p.grad.flatten().double() for p in m.parameters()]), loss.item())
    _, hip = _pair('fastscnn')
    hip.train()
    out = hip(x.to(DEV))
    loss_h = tssa.cross_entropy(out, y.to(DEV), ignore_index=255)
    loss_h.backward()
    gh = torch.cat([p.grad.flatten().double().cpu() for p in hip.parameters()])
    g32, g64 = grads[torch.float32][0], grads[torch.float64][0]
    e_ref32 = ((g32 - g64).norm() / g64.norm()).item()
    e_hip = ((gh - g64).norm() / g64.norm()).item()
    print('config-1 FastSCNN gradients: |hip - f64| %.3e, |oracle f32 - f64| %.3e' % (e_hip, e_ref32))
    assert abs(loss_h.item() / grads[torch.float64][1] - 1) < 1e-5
    assert e_hip <= max(2e-3, 3 * e_ref32), (e_hip, e_ref32)


def test_config1_contextnet14_train_step_gradients_vs_f64_oracle():
    """ContextNet14 had no whole-gradient check at config 1.  Forward + CE + backward in f32 against the f64
    oracle: the loss to 1e-5; the whole 1.1 M-element gradient within 3x the f32 oracle's own distance from f64 (its 40-layer
    context branch amplifies rounding: that distance is ~0.1 and is printed -- the yardstick, not a tolerance we chose); and the
    part of the gradient that does NOT pass through the context branch (classifier, fusion module, spatial branch: the
    well-conditioned tensors) under an absolute 5e-3."""
    import torch_semantic_segmentation_amd as tssa
    x, y = synthetic_batch(4, 512, 1024)
    loss_fn = nn.CrossEntropyLoss(ignore_index=255)
    grads = {}
    for dt in (torch.float32, torch.float64):
        torch.manual_seed(0)
        m = O.build('contextnet14')
        cases.zero_dropout(m)
        m.to(dt).train()
        loss = loss_fn(m(x.to(dt)), y)
        loss.backward()
        grads[dt] = ({n: p.grad.double() for n, p in m.named_parameters()}, loss.item())
    _, hip = _pair('contextnet14')
    hip.train()
    loss_h = tssa.cross_entropy(hip(x.to(DEV)), y.to(DEV), ignore_index=255)
    loss_h.backward()
    gh = {n: p.grad.double().cpu() for n, p in hip.named_parameters()}
    g32, g64 = grads[torch.float32][0], grads[torch.float64][0]
    cat = lambda d, keys: torch.cat([d[k].flatten() for k in keys])      # noqa: E731
    allk = list(g64)
    e_ref32 = ((cat(g32, allk) - cat(g64, allk)).norm() / cat(g64, allk).norm()).item()
    e_hip = ((cat(gh, allk) - cat(g64, allk)).norm() / cat(g64, allk).norm()).item()
    good = [k for k in allk if k.startswith(('classifier.', 'feature_fusion.', 'spatial.'))]
    e_ref32_good = ((cat(g32, good) - cat(g64, good)).norm() / cat(g64, good).norm()).item()
    e_hip_good = ((cat(gh, good) - cat(g64, good)).norm() / cat(g64, good).norm()).item()
    print('config-1 ContextNet14 gradients: whole |hip - f64| %.3e (oracle f32: %.3e); without the context branch (%d tensors) '
          '|hip - f64| %.3e (oracle f32: %.3e)' % (e_hip, e_ref32, len(good), e_hip_good, e_ref32_good))
    assert abs(loss_h.item() / grads[torch.float64][1] - 1) < 1e-5
    assert e_hip <= max(2e-3, 3 * e_ref32), (e_hip, e_ref32)
    assert e_hip_good <= max(5e-3, 3 * e_ref32_good), (e_hip_good, e_ref32_good)


def test_config2_full_size_f32_train_step_vs_cpu_oracle():
    """the full-size anchor.  BASELINE config 2's workload, 8 x 3 x 1024 x 2048, in f32 through the general
    kernels (the path the bf16 full-size test uses as ITS reference) against the CPU oracle itself, one step: the loss to 1e-5
    of the f64 oracle, and every `classifier.*` gradient (the well-conditioned end of the network) against the f64 oracle within
    max(1e-3, 3 x the f32 oracle's own distance from f64) -- two f32 implementations of a 16.8 M-pixel sum differ by ~1e-3 from
    each other, measured here and printed, so the yardstick is the reference's own rounding, not a number we picked.  BatchNorm
    biases in front of (linear conv -> BatchNorm) have an analytically zero gradient: errors are measured against at least 1 %
    of the largest gradient RMS of the same kind.  ~50 GB of host memory and ~60 s of CPU time for the oracle's two steps."""
    import torch_semantic_segmentation_amd as tssa
    x, y = synthetic_batch(8, 1024, 2048)
    grads = {}
    for dt in (torch.float32, torch.float64):
        torch.manual_seed(0)
        m = O.build('fastscnn')
        cases.zero_dropout(m)
        m.to(dt).train()
        loss = nn.functional.cross_entropy(m(x.to(dt)), y, ignore_index=255)
        loss.backward()
        grads[dt] = ({n: p.grad.double().clone() for n, p in m.named_parameters() if n.startswith('classifier.')}, loss.item())
        del m, loss
    _, hip = _pair('fastscnn')
    hip.train()
    loss_h = tssa.cross_entropy(hip(x.to(DEV)), y.to(DEV), ignore_index=255)
    loss_h.backward()
    g32, g64 = grads[torch.float32][0], grads[torch.float64][0]
    assert abs(loss_h.item() / grads[torch.float64][1] - 1) < 1e-5, (loss_h.item(), grads[torch.float64][1])
    kind = lambda n, t: 'w' if t.dim() == 4 else ('gamma' if n.endswith('weight') else 'beta')      # noqa: E731
    top = {}
    for n, t in g64.items():
        top[kind(n, t)] = max(top.get(kind(n, t), 0.0), (t.norm() / t.numel() ** 0.5).item())
    rows, bad = [], []
    for n, p in hip.named_parameters():
        if n in g64:
            den = max(g64[n].norm().item(), 1e-2 * top[kind(n, g64[n])] * g64[n].numel() ** 0.5)
            e_hip = ((p.grad.double().cpu() - g64[n]).norm() / den).item()
            e_ref = ((g32[n] - g64[n]).norm() / den).item()
            rows.append((e_hip, e_ref, n))
            if e_hip > max(1e-3, 3 * e_ref):
                bad.append((n, e_hip, e_ref))
    rows.sort(reverse=True)
    print('config-2 (8 x 3 x 1024 x 2048) f32 vs the f64 CPU oracle: loss %.6f vs %.6f; classifier gradients (hip, oracle-f32), worst first: %s'
          % (loss_h.item(), grads[torch.float64][1], ['%s %.2e %.2e' % (n, a, b) for a, b, n in rows[:6]]))
    assert len(rows) >= 10 and not bad, bad


# Parameters whose gradients are WELL CONDITIONED at this size: everything downstream of the last tiny-sample BatchNorm.
# FastSCNN's pyramid arm with bin 1 normalises over B = 8 values per channel, ContextNet's context branch is a 40-layer
# chain of batch-statistics BatchNorms on 1/32-1/128 maps: gradients that flow back through them amplify rounding by
# ~1e5 (the reference's own f32 run is 1e-2 away from its f64 run at config 1, tests above), so with bf16 storage
# (2^-9) they decorrelate completely -- for ANY implementation.  The tensors listed here receive their gradient from the
# loss through the decoder only; they include the largest layers of the step (128 channels at 1/8 resolution).
WELL_CONDITIONED = {'fastscnn': ('classifier.', 'fusion.', 'features.3.conv.'),
                    'contextnet14': ('classifier.', 'feature_fusion.', 'spatial.')}


# Even these drift apart in bf16, layer by layer going backwards (measured, round 2: last conv 2-5e-3, one block earlier
# 5-10e-2, the fusion module 0.3-0.5; the general and the lean bf16 kernels differ from EACH OTHER by half as much): with
# random-init weights, N(0,1) images and random targets the gradient is the sqrt(N) residual of 16.8 M cancelling pixel
# terms, every ReLU layer flips ~0.3 % of its masks under a 2^-9 perturbation and every BatchNorm backward subtracts the
# two dominant components.  So the absolute caps apply to the LAST layer only (TIGHT); for every tensor in the list the
# lean kernels must be no further from f32 than twice the general bf16 kernels, with a gradient norm within 2x of f32.
# The bf16 kernels themselves are pinned at these sizes by tests/test_gpu_lean_vs_oracle.py::test_baseline_sized_layers...
TIGHT = {'fastscnn': ('classifier.3.', 'classifier.1.3.'), 'contextnet14': ('classifier.5.', 'classifier.3.1.')}
CAPS = (1.5e-2, 1e-2)    # |lean - f32|, |lean - general| on the TIGHT tensors


@pytest.mark.parametrize('name', ['fastscnn', 'contextnet14'])
def test_baseline_size_bf16_train_step_through_the_graphed_trainer(name):
    """8 x 3 x 1024 x 2048, the benchmark's exact path: Trainer(use_graph=True) + FlatAdamW + fused head/loss + lean bf16
    kernels.  Reference at this size = the f32 general-kernel path of the library (pinned to the oracle by every 1e-3
    test of this suite at smaller sizes and by the config-1 tests above); yardstick = the general bf16 kernels on the
    same batch.  Loss to 1e-3; per-tensor gradients of the decoder-side tensors within 2x the yardstick, the last layer's
    under 1.5e-2 (and lean == general to 1e-2 there); every gradient finite and non-zero; running statistics;
    num_batches_tracked == 1 after one replay."""
    import torch_semantic_segmentation_amd as tssa
    from torch_semantic_segmentation_amd import engine as E
    from torch_semantic_segmentation_amd import _native as N
    torch.manual_seed(0)
    base = cases.product_model(name)
    cases.zero_dropout(base)
    state = copy.deepcopy(base.state_dict())
    names = [n for n, _ in base.named_parameters()]
    x, y = synthetic_batch(8, 1024, 2048)
    x, y = x.to(DEV), y.to(DEV)

    def run(dtype, disable_fast, graph):
        m = cases.product_model(name)
        m.load_state_dict(state, strict=True)
        cases.zero_dropout(m)
        m.to(DEV)
        tssa.set_compute_dtype(m, dtype)
        opt = E.FlatAdamW(m.parameters(), lr=1e-3, weight_decay=1e-5)
        tr = E.Trainer(m, opt, tssa.CrossEntropyLoss(ignore_index=255), use_graph=graph)
        N.call('tss_set_option', 1, int(disable_fast))
        try:
            loss = tr.step_async(x, y).item()
            torch.cuda.synchronize()
        finally:
            N.call('tss_set_option', 1, 0)
        nbt = [int(b) for k, b in m.named_buffers() if k.endswith('num_batches_tracked')]
        stats = torch.cat([b.detach().flatten().double() for k, b in m.named_buffers() if k.endswith(('running_mean', 'running_var'))]).cpu()
        grads = [p.grad.detach().double().cpu().clone() for p in m.parameters()]
        del tr, opt, m
        torch.cuda.empty_cache()
        return loss, grads, nbt, stats
    l32, g32, _, s32 = run(torch.float32, True, False)
    lgen, ggen, _, sgen = run(torch.bfloat16, True, False)
    llean, glean, nbt, slean = run(torch.bfloat16, False, True)
    assert all(v == 1 for v in nbt), 'num_batches_tracked must be 1 after one replayed step (warm-up runs leave no trace)'
    assert all(torch.isfinite(t).all() and t.norm() > 0 for t in glean), 'every parameter gets a finite, non-zero gradient'
    print('%s 8x3x1024x2048: loss f32 %.6f  bf16-general %.6f  bf16-lean(graph) %.6f' % (name, l32, lgen, llean))
    assert abs(llean / l32 - 1) < 1e-3 and abs(lgen / l32 - 1) < 1e-3
    rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-300)).item()   # noqa: E731
    worst, bad = [], []
    for n, a32, agen, alean in zip(names, g32, ggen, glean):
        if n.startswith(WELL_CONDITIONED[name]) and a32.norm() > 1e-6:
            e_gen, e_lean, e_dir = rel(agen, a32), rel(alean, a32), rel(alean, agen)
            ratio = (alean.norm() / a32.norm()).item()
            worst.append((e_lean, e_gen, e_dir, ratio, n))
            ok = e_lean <= 2 * e_gen + 2e-3 and 0.5 < ratio < 2.0
            if n.startswith(TIGHT[name]):
                ok = ok and e_lean <= CAPS[0] and e_dir <= CAPS[1]
            if not ok:
                bad.append((n, e_lean, e_gen, e_dir, ratio))
    worst.sort(reverse=True)
    print('   %d well-conditioned tensors (lean vs f32, general vs f32, lean vs general, norm ratio):' % len(worst))
    for e_lean, e_gen, e_dir, ratio, n in worst:
        print('     %-40s %.3e %.3e %.3e %.3f' % (n, e_lean, e_gen, e_dir, ratio))
    assert not bad, bad
    assert len(worst) >= 20
    assert ((slean - s32).norm() / s32.norm()).item() <= 2 * ((sgen - s32).norm() / s32.norm()).item() + 1e-3


def test_config5_eval_2048x4096_vs_cpu_oracle_and_fused_head():
    """1 x 3 x 2048 x 4096 eval forward (row-sliced pyramid pooling, 1000+ tile sweeps, the x8 head on 8.4 M pixels):
    f32 logits within 1e-3 of the CPU oracle, argmax exact outside sub-resolution ties; the fused upsample + argmax +
    confusion operator equals argmax_confusion on the materialised logits; bf16 lean forward within the bf16 yardstick."""
    import torch_semantic_segmentation_amd as tssa
    from torch_semantic_segmentation_amd import _native as N
    ref, hip = _pair('fastscnn')
    x, y = synthetic_batch(1, 2048, 4096)
    ref.eval(); hip.eval()
    with torch.no_grad():
        want = ref(x)
        got = hip(x.to(DEV))
        pred, cm = tssa.argmax_confusion(got, y.to(DEV), ignore_index=255)
        low = hip.forward_lowres(x.to(DEV))
        pred_f, cm_f = tssa.upsample_argmax_confusion(low, y.to(DEV), scale_factor=8, ignore_index=255)
    rel = cases.rel_err(got.cpu().numpy(), want.numpy())
    assert rel < 1e-3, rel
    frac = _argmax_check(got, want, pred)
    # fused head: same arithmetic order is not guaranteed -> allow flips only at unresolvable ties
    mism = pred_f != pred
    top2 = got.topk(2, dim=1).values
    gap = (top2[:, 0] - top2[:, 1])
    assert (gap[mism] < 1e-5 * got.abs().max()).all()
    assert int(cm.sum()) == int((y != 255).sum()) == int(cm_f.sum())
    assert (cm - cm_f).abs().sum().item() <= 2 * int(mism.sum())
    print('config-5 f32: logits rel err %.2e, argmax mismatch vs oracle %.2e, fused-vs-unfused flips %d' % (rel, frac, int(mism.sum())))
    with torch.no_grad():
        errs = {}
        for disable in (True, False):
            tssa.set_compute_dtype(hip, torch.bfloat16)
            N.call('tss_set_option', 1, int(disable))
            try:
                lb = hip.forward_lowres(x.to(DEV)).float()
            finally:
                N.call('tss_set_option', 1, 0)
            errs[disable] = ((lb - low).norm() / low.norm()).item()
        tssa.set_compute_dtype(hip, torch.float32)
    print('config-5 bf16 low-res logits L2 err vs f32: general %.3e lean %.3e' % (errs[True], errs[False]))
    assert errs[False] <= 2 * errs[True] + 2e-3
