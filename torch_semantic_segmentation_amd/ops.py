"""Autograd operators of the HIP path: thin Python glue over include/tss_hip.h.

Design: a convolution stores its output RAW.  The BatchNorm(+ReLU) that follows it in the
reference's blocks (TSS/models/fastscnn.py:164-199, TSS/models/contextnet.py:150-177) is carried as a
`Deferred` (raw tensor + per-channel affine + relu flag) and applied by the consumer while it loads its
input; a normalised activation is only written where the reference needs a real tensor (block outputs,
residual sums) by `join`.  Backward mirrors it: between two of these operators the tensor handed back
through autograd for a raw conv output is e = d(loss)/d(BN output), already ReLU-masked, and the
BatchNorm backward (which needs two per-channel sums over e) is applied on load by the producer's
backward kernels from coefficients finalised out of `BNLink.bstats`.  Tensors that leave a block are
ordinary tensors with ordinary gradients.

There is no torch/ATen fallback here: every operator calls the C ABI and raises if it is unavailable.
"""
import ctypes
import os
import threading

import torch
from torch.autograd import Function
from torch.nn.modules.batchnorm import _BatchNorm

from . import _native as N

call, ptr, stream = N.call, N.ptr, N.stream


# ----------------------------------------------------------------------------- direct gradient accumulation

_DIRECT = [False]


class direct_grads:
    """While active, parameter gradients are accumulated by the kernels straight into an existing float32
    `.grad` (e.g. FlatAdamW's flat buffer) and autograd is handed None for them: no per-parameter zero-fill
    and no `.grad +=` launch.  Off by default: then every backward returns ordinary gradient tensors."""

    def __init__(self, enabled=True):
        self.enabled = bool(enabled)

    def __enter__(self):
        self.prev = _DIRECT[0]
        _DIRECT[0] = self.enabled
        return self

    def __exit__(self, *exc):
        _DIRECT[0] = self.prev
        return False


# Launch a layer's weight gradient on a side stream, concurrently with its input gradient (ConvUnitFn.backward).
overlap_wgrad = os.environ.get('TSS_OVERLAP_WGRAD', '0') == '1'   # measured: no gain on MI355X once the kernels are pipelined (9.88 ms off vs 9.84-9.95 ms on)
# depthwise layers: input gradient and weight gradient from one kernel (tss_dwconv3x3_bwd_fused) instead of two launches.
# Opt-in: correct, reads e / y / x once, but measured slower on MI355X (1 wave/SIMD, see dwconv.hip): 7.28 vs 7.05 ms/step.
fuse_dw_backward = os.environ.get('TSS_FUSE_DW_BWD', '0') == '1'
overlap_max_elems = int(os.environ.get('TSS_OVERLAP_MAX', str(48 << 20)))   # only layers too small to fill the chip on their own (large ones just contend)
overlap_min_elems = int(os.environ.get('TSS_OVERLAP_MIN', '0'))
# the row reductions of the one-sweep depthwise backward, collected over a backward pass and summed in one launch at its end
# (only for gradients written straight into their final buffer: a gradient RETURNED to autograd must be complete on return)
batch_dw_reductions = os.environ.get('TSS_BATCH_DW_REDUCE', '1') == '1'
fuse_pw_backward = os.environ.get('TSS_PW_BWD_FUSED', '1') != '0'     # csrc/pwbwd.hip, for the layers it prefers
sweep_pw_backward = os.environ.get('TSS_PW_SWEEP', '1') != '0'        # csrc/pwsweep.hip: the large 1x1 layers, one sweep (round 4)
# Backward-pass scheduling ("launch count"): every dependent launch costs >= 4.7 us in the replayed step, so
#   * the weight gradient of a 1x1 layer -- which nothing downstream in the backward pass depends on -- is POSTPONED until the
#     next BatchNorm-backward finalize is due, and that finalize rides in front of its grid (tss_pwconv_bwd_weight's `fin`);
#   * the slot reduction of a launched weight gradient rides in front of the next backward-data grid of ANY 1x1 layer.
# Everything left over is flushed by the end-of-pass callback, so gradients are complete when backward() returns (and not before:
# see direct_grads).
postpone_wgrad = os.environ.get('TSS_POSTPONE_WGRAD', '1') != '0'
# weight gradient of the three-tap (1x3 / 3x1) layers: unfold + the pipelined pointwise kernel (0: the generic tap-loop kernel)
unfold_1d_wgrad = os.environ.get('TSS_FC1D_WGRAD', '1') != '0'
_CONST_ROWS = {}


def _const_rows(dev, n):
    """(ones, zeros) f32 rows of at least n channels on `dev`: BatchNorm-backward coefficients of a layer that has none."""
    key = (dev.type, dev.index)
    got = _CONST_ROWS.get(key)
    if got is None or got[0].numel() < n:
        m = max(768, n)
        got = _CONST_ROWS[key] = (torch.ones(m, dtype=torch.float32, device=dev), torch.zeros(m, dtype=torch.float32, device=dev))
    return got
# Layers that run inside a two-stream region of the forward pass (ContextNet's branches, `overlap_region`) are not postponed: there
# the other stream's kernels already fill the launch gaps, and what postponing leaves for the end of the pass (a weight gradient and
# its slot reduction per stream, un-overlapped) costs more than the finalize launches it saves (measured: 6.32 vs 6.21 ms per step).
_overlap_depth = [0]
postpone_in_overlap = os.environ.get('TSS_POSTPONE_OVERLAPPED', '0') == '1'     # A/B: postpone inside two-stream regions too


class _Pass:
    """Scheduling state of ONE running backward pass (one autograd graph task).  Round 4: the state used to be module-global and
    was cleared whenever the graph-task id changed, so a nested pass (torch.utils.checkpoint(use_reentrant=True), a hook calling
    torch.autograd.grad) silently deleted the OUTER pass's postponed weight gradients and row reductions.  Now every graph task has
    its own: a nested pass postpones, flushes and completes its own work, the outer one keeps its own.
    `wg` / `red` are keyed by the stream the postponed work belongs to: a model whose branches run on two streams (ContextNet) has
    two independent backward chains, and a launch postponed on one stream must neither ride on nor carry work of the other."""
    __slots__ = ('dw', 'wg', 'red')

    def __init__(self):
        self.dw = []        # (ws, dw, ncols, nrows): row reductions for the one launch at the end of the pass
        self.wg = {}        # stream id -> (launch(fin_job or None), device, stream)
        self.red = {}       # stream id -> (ws, dw, P, K, N, device, stream)


_passes = {}                # graph-task id -> _Pass
_lock = threading.RLock()   # two trainers in two threads (each with its own autograd device thread) share this module


class BnBwdJob(ctypes.Structure):
    """tss_bn_bwd_job of include/tss_hip.h."""
    _fields_ = [('bstats', ctypes.c_void_p), ('count', ctypes.c_double), ('invstd', ctypes.c_void_p), ('gamma', ctypes.c_void_p),
                ('training', ctypes.c_int), ('accumulate', ctypes.c_int), ('dgamma', ctypes.c_void_p), ('dbeta', ctypes.c_void_p),
                ('ga', ctypes.c_void_p), ('gb', ctypes.c_void_p), ('gce', ctypes.c_void_p), ('C', ctypes.c_int),
                ('xchg_world', ctypes.c_int), ('xchg_rank', ctypes.c_int), ('xchg_peers', ctypes.c_void_p * 8), ('xchg_counters', ctypes.c_void_p)]


def _backward_task():
    """Id of the running autograd graph task (-1: none, or nothing can be deferred).  The first call of a pass creates its state and
    queues its end-of-pass flush on THAT graph task."""
    try:
        task = torch._C._current_graph_task_id()
    except AttributeError:          # private API gone: nothing is deferred
        return -1
    if task == -1:
        return -1
    with _lock:
        if task not in _passes:
            try:
                torch.autograd.variable.Variable._execution_engine.queue_callback(lambda _t=task: _end_of_backward(_t))
            except Exception:           # noqa: BLE001 -- no callback, no deferral
                return -1
            # state of passes that raised before their callback ran is void; graph-task ids only grow, and a running outer pass of a
            # nested one is never more than a few ids behind: keep the newest 16
            for old in sorted(_passes)[:-15]:
                del _passes[old]
            _passes[task] = _Pass()
    return task


def _cur_pass():
    task = _backward_task()
    return _passes.get(task) if task != -1 else None


def _flush_wg(sid=None, ps=None):
    """Launch postponed weight gradients now, without a rider, each on the stream it belongs to (sid: only that stream's)."""
    ps = ps if ps is not None else _cur_pass()
    if ps is None:
        return
    for k in ([sid] if sid is not None else list(ps.wg)):
        with _lock:
            ent = ps.wg.pop(k, None)
        if ent is not None:
            launch, dev, st_ = ent
            with torch.cuda.device(dev), torch.cuda.stream(st_):
                launch(None)


def _flush_red(sid=None, ps=None):
    ps = ps if ps is not None else _cur_pass()
    if ps is None:
        return
    for k in ([sid] if sid is not None else list(ps.red)):
        with _lock:
            ent = ps.red.pop(k, None)
        if ent is not None:
            ws, dw, P, K, Nn, dev, st_ = ent
            with torch.cuda.device(dev), torch.cuda.stream(st_):
                call('tss_pwconv_wg_reduce', ptr(ws), ptr(dw), P, K, Nn, stream())


def _take_red(dev):
    """(ws ptr, dw ptr, P, K, N) of a slot reduction waiting for a carrier on THIS stream (else NULLs) + its tensors (kept alive by the caller)."""
    ps = _cur_pass()
    ent = None
    if ps is not None:
        with _lock:
            ent = ps.red.pop(stream(), None)
    if ent is not None:
        ws, dw, P, K, Nn = ent[:5]
        return (ptr(ws), ptr(dw), P, K, Nn), (ws, dw)
    return (None, None, 0, 0, 0), None


def _end_of_backward(task):
    with _lock:
        ps = _passes.pop(task, None)
    if ps is None:
        return
    _flush_wg(ps=ps)
    _flush_red(ps=ps)
    # the row reductions below run on the current stream: order it behind the side streams whose kernels wrote some of the rows
    # (the autograd engine joins its leaf streams only AFTER the final callbacks)
    if ps.dw and _side_streams:
        cur = torch.cuda.current_stream()
        for s_ in _side_streams.values():
            if s_.device == cur.device:
                cur.wait_stream(s_)
    _flush_dw_reductions(ps)


def _bn_bwd_finalize(link, C, acc, dgamma, dbeta, st, training=None):
    """BatchNorm-backward coefficients + d(gamma), d(beta) of `link`: in front of a postponed weight-gradient grid of the same
    stream (and the same pass) when there is one (no launch of its own), else tss_bn_bwd_finalize.  A cross-replica BatchNorm
    (link.sync): the sums cross the ranks inside the same finalize blocks, riding or not (IPC exchange, csrc/xchg.hip) -- or, without
    the exchange, through the process group's all-reduce."""
    training = int(link.training if training is None else training)
    ex = None
    if link.sync is not None:
        ex = _exchange(link.sync, link.bstats.device) if C <= 768 else None
        if ex is None:
            _flush_wg(st)
            gs = _allreduce_stats(link.bstats, link.count, C, link.sync, st)
            call('tss_bn_bwd_finalize_sync', ptr(link.bstats), ptr(gs), ptr(link.invstd), ptr(link.gamma), int(acc),
                 ptr(dgamma), ptr(dbeta), ptr(link.ga), ptr(link.gb), ptr(link.gce), C, st)
            return
    ps = _cur_pass()
    ent = None
    if ps is not None:
        with _lock:
            ent = ps.wg.get(st)
            if ent is not None and ent[1] == link.bstats.device:
                ps.wg.pop(st, None)
            else:
                ent = None
    if ent is not None:
        job = BnBwdJob(ptr(link.bstats), float(link.count), ptr(link.invstd), ptr(link.gamma), training, int(acc),
                       ptr(dgamma), ptr(dbeta), ptr(link.ga), ptr(link.gb), ptr(link.gce), int(C))
        if ex is not None:
            job.xchg_world, job.xchg_rank, job.xchg_counters = ex.world, ex.rank, ptr(ex.counters)
            for r_ in range(ex.world):
                job.xchg_peers[r_] = ex.peers[r_]
        ent[0](job)
        return
    if ex is not None:
        call('tss_bn_bwd_finalize_xchg', ptr(link.bstats), float(link.count), ex.peers, ex.rank, ex.world, ptr(ex.counters),
             ptr(link.invstd), ptr(link.gamma), int(acc), ptr(dgamma), ptr(dbeta), ptr(link.ga), ptr(link.gb), ptr(link.gce), C, st)
        return
    call('tss_bn_bwd_finalize', ptr(link.bstats), float(link.count), ptr(link.invstd), ptr(link.gamma), training, int(acc),
         ptr(dgamma), ptr(dbeta), ptr(link.ga), ptr(link.gb), ptr(link.gce), C, st)


def _flush_dw_reductions(ps):
    with _lock:
        jobs = list(ps.dw)
        del ps.dw[:]
    if not jobs:
        return
    n = len(jobs)
    ws = (ctypes.c_void_p * n)(*[j[0].data_ptr() for j in jobs])
    dw = (ctypes.c_void_p * n)(*[j[1].data_ptr() for j in jobs])
    cols = (ctypes.c_int * n)(*[j[2] for j in jobs])
    rows = (ctypes.c_int * n)(*[j[3] for j in jobs])
    with torch.cuda.device(jobs[0][0].device):
        call('tss_dw_reduce_many', n, ws, dw, cols, rows, stream())


def _reduce_rows_now(ws, dw, ncols, nrows):
    a, b = (ctypes.c_void_p * 1)(ws.data_ptr()), (ctypes.c_void_p * 1)(dw.data_ptr())
    call('tss_dw_reduce_many', 1, a, b, (ctypes.c_int * 1)(ncols), (ctypes.c_int * 1)(nrows), stream())


def _param_observed(p):
    """A parameter whose gradient somebody may read BEFORE backward() returns (tensor hooks, post-accumulate-grad hooks, e.g. a
    bucketed all-reduce that overlaps the backward pass): its weight gradient is completed in place, never deferred."""
    return p is not None and bool(getattr(p, '_backward_hooks', None) or getattr(p, '_post_accumulate_grad_hooks', None))


def _defer_dw_reduction(ws, dw, ncols, nrows, param=None):
    """Queue the row reduction of a one-sweep backward for the single launch at the end of the pass (now, if nothing can be deferred).
    Contract of the deferral (and of the postponed 1x1 weight gradients): directly accumulated gradients are complete when backward()
    returns, not earlier -- except for parameters with hooks, which are finished at once."""
    ps = _cur_pass()
    if ps is None or _param_observed(param):
        _reduce_rows_now(ws, dw, ncols, nrows)
        return
    if ps.dw and ps.dw[0][0].device != ws.device:
        _flush_dw_reductions(ps)
    with _lock:
        ps.dw.append((ws, dw, ncols, nrows))


class _PerThread:
    """A dict private to the calling thread.  The forward-pass registries below (keyed by id() of a tensor between the operator that
    registers an entry and the operator that picks it up, both in the same forward call) are per thread, so two models running their
    forward passes in two threads never see -- or steal -- each other's entries."""

    def __init__(self):
        self._tl = threading.local()

    def _d(self):
        d = getattr(self._tl, 'd', None)
        if d is None:
            d = self._tl.d = {}
        return d

    def __bool__(self):
        return bool(self._d())

    def __len__(self):
        return len(self._d())

    def __setitem__(self, k, v):
        self._d()[k] = v

    def __contains__(self, k):
        return k in self._d()

    def pop(self, k, default=None):
        return self._d().pop(k, default)

    def get(self, k, default=None):
        return self._d().get(k, default)

    def clear(self):
        self._d().clear()


# Residual blocks: the block input has two consumers (the first 1x1 layer and the skip), so its gradient is a sum of two tensors.
# Autograd would add them with an elementwise launch of its own; instead the skip's gradient (produced first, by the join's
# backward) is handed to the first layer's backward-data kernel, which adds it in its epilogue (tss_pwconv_bwd_data_radd).
fold_residual_adds = os.environ.get('TSS_FOLD_RESIDUAL', '1') != '0'
_pending_forks = _PerThread()


class _Fork:
    __slots__ = ('g2', 'consumed', 'prev_join')

    def __init__(self):
        self.g2, self.consumed, self.prev_join = None, False, None


class ForkFn(Function):
    """x -> (x, x): two handles of one tensor whose gradients meet here.  If the consumer of the first handle has already added the
    second handle's gradient into its own (fork.consumed), that sum is passed on as it is."""

    @staticmethod
    def forward(ctx, x, fork):
        ctx.fork = fork
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        fork = ctx.fork
        consumed = fork.consumed
        fork.g2, fork.consumed = None, False
        if ga is None:
            return gb, None
        if gb is None or consumed:
            return ga, None
        return ga + gb, None


def residual_fork(x):
    """(handle for the convolution path, handle for the skip, fork) of a residual block's input; (x, x, None) when nothing is folded."""
    if not (fold_residual_adds and torch.is_grad_enabled() and torch.is_tensor(x) and x.requires_grad and x.is_cuda
            and x.dtype == torch.bfloat16 and not N.fast_paths_disabled()):
        return x, x, None
    fork = _Fork()
    fork.prev_join = _take_join(x)         # x is a block output: its join's backward can ride in the first layer's backward-data launch
    xa, xb = ForkFn.apply(x, fork)
    _pending_forks[id(xa)] = fork          # picked up by the first conv unit that consumes xa (conv_unit)
    return xa, xb, fork


# A relu join whose output feeds a 1x1 layer (block output -> the next block's expand conv): the join's backward (ReLU mask of its
# output + the BatchNorm-backward sums) can run in the epilogue of that layer's backward-data launch (tss_pwconv_bwd_data_joined).
# join() registers its configuration here, keyed by the output tensor; the consuming conv unit picks it up.
fuse_join_backward = os.environ.get('TSS_FUSE_JOIN_BWD', '1') != '0'
# ... for block outputs up to this many elements: the two extra loads of the epilogue drain the next tile's prefetch, which costs a long
# layer more than the join launch it saves (262 k pixels x 64 channels: +48 us against a 28 us join; 65 k x 64: +10 against 14;
# 16 k x 96-128: +2 against 9)
fuse_join_backward_max = int(os.environ.get('TSS_FUSE_JOIN_BWD_MAX', '4000000'))
_join_ctx = _PerThread()


def _take_join(x):
    if not _join_ctx or not torch.is_tensor(x):
        return None
    ref = _join_ctx.pop(id(x), None)
    cfg = ref() if ref is not None else None
    return cfg if (cfg is not None and cfg.j_ptr == x.data_ptr()) else None


_pending_stash = _PerThread()


def fork_two(x):
    """(xa, xb, fork) for a materialised activation with TWO consumers whose gradients would otherwise meet in an elementwise launch of
    autograd's: the consumer of xb (created later, so its backward runs first) leaves its input gradient in fork.g2, the 1x1 layer
    that consumes xa adds it in the epilogue of its backward-data kernel (tss_pwconv_bwd_data_radd) -- or, for pooled maps,
    tss_ppm_pool_bwd does.  Falls back to autograd's add whenever a side cannot take part.  (x, x, None) when nothing is folded."""
    xa, xb, fork = residual_fork(x)
    if fork is not None:
        _pending_stash[id(xb)] = fork          # picked up by the operator that consumes xb
    return xa, xb, fork


def drop_fork(xa, xb):
    """Forget fork registrations nobody picked up (a consumer outside the envelope): ids may be recycled."""
    _pending_forks.pop(id(xa), None)
    _pending_stash.pop(id(xb), None)


# independent branches of a model (ContextNet's spatial / context branches) on two streams: parallel branches of the captured graph
overlap_branches = os.environ.get('TSS_OVERLAP_BRANCHES', '1') != '0'     # measured: ContextNet14 step 6.34 -> 6.27 ms


class overlap_region:
    """Marks the part of a forward pass whose layers run on two streams at once (see _overlap_depth)."""

    def __enter__(self):
        _overlap_depth[0] += 1
        return self

    def __exit__(self, *exc):
        _overlap_depth[0] -= 1
        return False


def tensors_of(x):
    """The device tensors behind an activation (a tensor, or a Deferred with its BatchNorm link)."""
    if isinstance(x, Deferred):
        out = [x.raw]
        if x.link is not None:
            out += [x.link.vec, x.link.stats]
        return out
    return [x] if torch.is_tensor(x) else []


_side_streams = {}


def _side_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]



# bf16 shadows of the 1x1 convolution weights (WeightShadows below): {weight.data_ptr(): (copy [N][K], transpose [K][N])},
# only populated while a `with shadows:` block is active, i.e. while somebody guarantees they are current.
_SHADOWS = {}


class WeightShadows:
    """bf16 copies (and transposes) of every 1x1 convolution weight of a model, refreshed by ONE kernel launch
    (`tss_cast_weights`).  The pointwise kernels then stage their weight tiles with plain 16-byte copies instead of
    converting f32 in every block.  Opt-in and explicit: the owner calls refresh() whenever the weights changed
    (Trainer: at the top of every step, inside the captured graph) and wraps the pass in `with shadows:`; outside such
    a block the kernels read the f32 weights as before, so a stale shadow can never be used."""

    def __init__(self, module):
        ws = [p for p in module.parameters() if p.dim() == 4 and p.shape[2] == 1 and p.shape[3] == 1
              and p.dtype == torch.float32 and p.is_cuda and p.shape[1] % 8 == 0]
        self.weights = ws
        self.table = None
        self.entries = {}
        if not ws:
            return
        dev = ws[0].device
        rows, off = [], 0
        for p in ws:
            n, k = p.shape[0], p.shape[1]
            rows.append((p, off, n, k))
            off += (n * k + 7) // 8 * 8                                # 16-byte aligned segments
        self.flat = torch.empty(off, dtype=torch.bfloat16, device=dev)
        self.flat_t = torch.empty(off, dtype=torch.bfloat16, device=dev)
        table = []
        for p, o, n, k in rows:
            c, t = self.flat[o:o + n * k], self.flat_t[o:o + n * k]
            self.entries[p.data_ptr()] = (c, t)
            table.append([p.data_ptr(), c.data_ptr(), t.data_ptr(), n, k])
        self.table = torch.tensor(table, dtype=torch.int64, device=dev)
        self.blocks = max(1, min(64, (max(n * k for _, _, n, k in rows) + 255) // 256))

    def refresh(self, zero=None):
        """Rewrite the shadows from the live weights.  `zero` (a contiguous float32 tensor, e.g. the flat gradient buffer) is
        cleared by the same launch.  Returns True when `zero` was cleared here."""
        if self.table is not None:
            for p in self.weights:                                        # parameters re-pointed since construction?
                if p.data_ptr() not in self.entries:
                    raise RuntimeError('WeightShadows: a parameter was reallocated; rebuild the shadows')
            fold = zero is not None and zero.dtype == torch.float32 and zero.is_contiguous() and zero.device == self.table.device
            call('tss_cast_weights', ptr(self.table), self.table.shape[0], self.blocks, ptr(zero) if fold else None,
                 zero.numel() if fold else 0, stream())
            return fold
        return False

    def __enter__(self):
        self.prev = dict(_SHADOWS)
        _SHADOWS.update(self.entries)
        return self

    def __exit__(self, *exc):
        _SHADOWS.clear()
        _SHADOWS.update(self.prev)
        return False


# bf16 tap-major copies of dense 3x3 weights prepared ahead of a forward (Dense3x3Shadows below): {weight.data_ptr(): [9][N][K] bf16}
_W3X3 = {}


class Dense3x3Shadows:
    """The bf16 [tap][output][input] copies of every dense 3x3 weight of a model that the matrix-core 3x3 kernels read (wstat.hip,
    atrous.hip, conv3x3.hip), written ahead of the forward instead of by one tss_permute_w3x3_bf16 launch per layer inside it.  Same
    contract as WeightShadows: refresh() rewrites them from the live weights; only visible inside `with shadows:`."""

    def __init__(self, module):
        self.items = []
        for m in module.modules():
            if (isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3) and m.groups == 1 and m.weight.is_cuda
                    and m.weight.dtype == torch.float32 and m.in_channels % 8 == 0 and m.out_channels % 8 == 0):
                w = m.weight
                self.items.append((w, w.data_ptr(), torch.empty((9, m.out_channels, m.in_channels), dtype=torch.bfloat16, device=w.device)))
        self.entries = {p: t for _, p, t in self.items}

    def refresh(self):
        for w, p, t in self.items:
            if w.data_ptr() != p:
                raise RuntimeError('Dense3x3Shadows: a parameter was reallocated; rebuild the shadows')
            call('tss_permute_w3x3_bf16', ptr(w), ptr(t), None, w.shape[0], w.shape[1], stream())

    def __enter__(self):
        self.prev = dict(_W3X3)
        _W3X3.update(self.entries)
        return self

    def __exit__(self, *exc):
        _W3X3.clear()
        _W3X3.update(self.prev)
        return False


# eval-mode BatchNorm affines computed for a whole model by one launch (EvalAffines below): {id(bn): [3][C] f32 view},
# only populated inside a `with affines:` block, for the forward that follows its refresh().
_EVAL_AFFINES = {}


class EvalAffines:
    """(mean, invstd, gamma*invstd) of every BatchNorm of a model that normalises with its running statistics, written
    by ONE kernel (`tss_bn_eval_affine_batched`) instead of one tiny launch per layer (44 in FastSCNN: 15 % of an
    eval-mode forward at 2048 x 4096).  Same contract as WeightShadows: the owner calls refresh() at the top of each
    forward (inside a captured graph it is replayed with it, so the values always follow the live parameters and
    buffers) and wraps the forward in `with affines:`; outside such a block every layer computes its own."""

    def __init__(self, module):
        import struct
        bns = [m for m in module.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)
               and m.running_mean is not None and m.running_var is not None and m.running_mean.is_cuda
               and m.running_mean.dtype == torch.float32 and (m.weight is None or m.weight.dtype == torch.float32)]
        self.bns, self.table, self.entries = bns, None, {}
        if not bns:
            return
        dev = bns[0].running_mean.device
        total = sum(3 * m.num_features for m in bns)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        rows, off = [], 0
        for m in bns:
            C = m.num_features
            out = self.flat[off:off + 3 * C].view(3, C)
            off += 3 * C
            self.entries[id(m)] = out
            eps_bits = struct.unpack('<i', struct.pack('<f', float(m.eps)))[0]
            rows.append([m.weight.data_ptr() if m.weight is not None else 0, m.running_mean.data_ptr(),
                         m.running_var.data_ptr(), out.data_ptr(), C, eps_bits])
        self.ptrs = [(r[0], r[1], r[2]) for r in rows]
        self.table = torch.tensor(rows, dtype=torch.int64, device=dev)
        self.max_c = max(m.num_features for m in bns)

    def refresh(self):
        if self.table is None:
            return
        for m, (g, rm, rv) in zip(self.bns, self.ptrs):                  # re-pointed parameters / buffers since construction?
            if (m.weight.data_ptr() if m.weight is not None else 0) != g or m.running_mean.data_ptr() != rm \
                    or m.running_var.data_ptr() != rv:
                raise RuntimeError('EvalAffines: a BatchNorm tensor was reallocated; rebuild the affines')
        call('tss_bn_eval_affine_batched', ptr(self.table), self.table.shape[0], self.max_c, stream())

    def __enter__(self):
        self.prev = dict(_EVAL_AFFINES)
        _EVAL_AFFINES.update(self.entries)
        return self

    def __exit__(self, *exc):
        _EVAL_AFFINES.clear()
        _EVAL_AFFINES.update(self.prev)
        return False


# The LDS-halo dense 3x3 kernel handles dilation <= 18, but it restages its nine 32 KB weight taps for every 64-pixel tile: on
# the 131 k-pixel map of BASELINE config 5 the three atrous branches of the ASPP head were 0.25 ms SLOWER through it than through
# the implicit-GEMM tap loop (3.41 vs 3.15 ms per image, profiles/README.md), so dilated convolutions stay on the general kernel
# unless this is raised (A/B: TSS_CONV3X3_LEAN_MAXDIL=18).
conv3x3_lean_max_dilation = int(os.environ.get('TSS_CONV3X3_LEAN_MAXDIL', '1'))


def _conv3x3_lean(dtype, k, n, stride, dil):
    """Domain of the LDS-halo dense 3x3 kernel (conv3x3.hip): contraction k, outputs n."""
    return (dtype == torch.bfloat16 and stride == 1 and 1 <= dil <= conv3x3_lean_max_dilation and k in (32, 64, 128) and n % 16 == 0 and 16 <= n <= 128
            and not N.fast_paths_disabled())


conv3x3_stream = os.environ.get('TSS_CONV3X3_STREAM', '1') != '0'


def _conv3x3_stream(dtype, k, n, stride, in_link, in_relu):
    """Domain of the register-streamed dense 3x3 kernel (atrous.hip): materialised bf16 input, stride 1, any dilation."""
    return (conv3x3_stream and dtype == torch.bfloat16 and stride == 1 and in_link is None and not in_relu and k % 32 == 0
            and 32 <= k <= 128 and n % 16 == 0 and 16 <= n <= 128 and not N.fast_paths_disabled())


def _shadow(weight, which):
    ent = _SHADOWS.get(weight.data_ptr()) if _SHADOWS else None
    return ptr(ent[which]) if ent is not None else None


def _aff(link):
    """(mean, scale, bias) device pointers of a pending BatchNorm: a = (x - mean) * scale + bias."""
    if link is None:
        return None, None, None
    return ptr(link.mean), ptr(link.scale), ptr(link.beta)


def _direct_target(param):
    g = param.grad if (param is not None and param.is_leaf) else None
    if _DIRECT[0] and g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.is_cuda:
        return g
    return None


# ----------------------------------------------------------------------------- layout helpers

def round_up(v, m):
    return (v + m - 1) // m * m


def new_nhwc(b, c, h, w, dtype, device, ld=None):
    """Logical (B,C,H,W) tensor stored NHWC with row pitch `ld` (>= C, multiple of 8)."""
    ld = round_up(c, 8) if ld is None else ld
    base = torch.empty((b, h, w, ld), dtype=dtype, device=device)
    t = base.permute(0, 3, 1, 2)
    return t if ld == c else t[:, :c]


def is_nhwc(t):
    if t.dim() != 4 or not t.is_cuda:
        return False
    b, c, h, w = t.shape
    ld = t.stride(3)
    return (t.stride() == (h * w * ld, 1, w * ld, ld) and ld % 8 == 0 and ld >= c
            and t.data_ptr() % 16 == 0)


def to_nhwc(t):
    """Canonical NHWC-strided view/copy of a logical NCHW tensor (the copy is boundary plumbing)."""
    if is_nhwc(t):
        return t
    _check_device(t)
    out = new_nhwc(*t.shape, t.dtype, t.device)
    out.copy_(t)
    return out


def _check_device(t):
    if not t.is_cuda:
        raise RuntimeError('torch_semantic_segmentation_amd runs on the MI355X HIP path only: got a %s tensor. '
                           'There is no CPU fallback (the CPU oracle lives in oracle/, for tests).' % t.device)
    if t.device.index is not None and t.device.index != torch.cuda.current_device():
        # the C ABI launches on the current device's current stream: a tensor of another device would be dereferenced there
        raise RuntimeError('tensor lives on cuda:%d but the current device is cuda:%d; wrap the call in '
                           'torch.cuda.device(%d) (one process per GPU is the supported layout)'
                           % (t.device.index, torch.cuda.current_device(), t.device.index))
    N.lib()


def ld(t):
    return t.stride(3)


def npix(t):
    return t.shape[0] * t.shape[2] * t.shape[3]


# ----------------------------------------------------------------------------- deferred BatchNorm

class BNLink:
    """State shared by the producer and the consumer of one deferred BatchNorm."""
    __slots__ = ('C', 'count', 'training', 'gamma', 'beta', 'vec', 'scale', 'mean', 'invstd',
                 'ga', 'gb', 'gce', 'stats', 'bstats', 'consumed', 'sync')

    def __init__(self, C, count, training, gamma, beta, device, slabs=True):
        self.C, self.count, self.training, self.gamma, self.beta = C, count, training, gamma, beta
        self.vec = torch.empty((6, C), dtype=torch.float32, device=device)
        self.mean, self.invstd, self.scale, self.ga, self.gb, self.gce = self.vec.unbind(0)
        if slabs:
            # [forward | backward][slab rows][sum, second moment]: written in full by the kernels, never cleared here
            both = torch.empty((2, N.stat_slabs(), 2 * C), dtype=torch.float64, device=device)
            self.stats, self.bstats = both[0], both[1]
        else:       # a unit that keeps its statistics on chip (ppm_arms): nobody writes slab rows for it
            self.stats = self.bstats = None
        self.consumed = False
        self.sync = None          # process group of a cross-replica (Sync) BatchNorm, see SyncBatchNorm below


class Deferred:
    """A raw conv output plus the BatchNorm(+ReLU) that its consumer still has to apply."""
    __slots__ = ('raw', 'link', 'relu')

    def __init__(self, raw, link=None, relu=False):
        self.raw, self.link, self.relu = raw, link, relu

    @property
    def shape(self):
        return self.raw.shape


    def take(self):
        """Mark the single allowed consumption of a BN-pending tensor (its backward sums are single-use)."""
        if self.link is not None:
            if self.link.consumed:
                raise RuntimeError('a deferred BatchNorm output may feed exactly one consumer; materialize it first')
            self.link.consumed = True
        return self


class SyncBatchNorm(torch.nn.BatchNorm2d):
    """Cross-replica BatchNorm for the data-parallel path (apex.parallel.SyncBatchNorm as the reference's scripts use it,
    scripts/train_fastscnn.py:144-145): in training mode the batch statistics are those of the GLOBAL batch.  Same
    parameters, buffers and state_dict keys as nn.BatchNorm2d.  The HIP path implements it as one all-reduce of a
    [2C+1] f64 vector per layer and direction between the producing kernel and the finalize kernel."""
    process_group = None


def convert_syncbn_model(module, process_group=None):
    """apex.parallel.convert_syncbn_model: every BatchNorm2d of `module` becomes a SyncBatchNorm, in place (the
    module objects, parameters and buffers stay the same, so optimizers and state_dicts are unaffected)."""
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm2d) and not isinstance(m, SyncBatchNorm):
            m.__class__ = SyncBatchNorm
            m.process_group = process_group
    return module


def _sync_group(bn, any_mode=False):
    """The process group over which this BatchNorm's statistics are reduced, or None (local statistics: not a Sync
    layer, eval mode unless `any_mode`, no process group, or a single rank)."""
    import torch.distributed as dist
    if not isinstance(bn, (SyncBatchNorm, torch.nn.SyncBatchNorm)) or not (bn.training or any_mode):
        return None
    if not (dist.is_available() and dist.is_initialized()):
        return None
    group = getattr(bn, 'process_group', None) or dist.group.WORLD
    # TSS_SYNCBN_FORCE=1: take the cross-replica path even in a one-rank group (exercises the collectives, their HIP-graph
    # capture and their cost on a single GPU; the result equals local BatchNorm)
    return group if (dist.get_world_size(group) > 1 or os.environ.get('TSS_SYNCBN_FORCE') == '1') else None


# ---- the exchange inside the finalize kernels (csrc/xchg.hip): mailboxes of the ranks of one node, mapped through HIP IPC
syncbn_ipc = os.environ.get('TSS_SYNCBN_IPC', '1') != '0'       # 0: every BatchNorm statistic through dist.all_reduce (RCCL), as in round 3
_XCHG = {}


class _Exchange:
    """This rank's mailbox + the peers' mailboxes as mapped into this process, for one (process group, device)."""

    def __init__(self, group, device):
        import socket
        import torch.distributed as dist
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = device
        if self.world > 8:
            raise RuntimeError('more than 8 ranks')
        nbytes = N.lib().tss_bn_xchg_bytes()
        own, handle = ctypes.c_void_p(), (ctypes.c_char * 64)()
        with torch.cuda.device(device):
            call('tss_ipc_alloc', nbytes, ctypes.byref(own), handle)
            self.own = own.value
            me = (socket.gethostname(), os.getpid(), bytes(handle.raw))
            everyone = [None] * self.world
            if self.world > 1:
                dist.all_gather_object(everyone, me, group=group)       # (also the barrier behind which every mailbox is zero-filled)
            else:
                everyone[0] = me
            if len({h for h, _, _ in everyone}) != 1:
                call('tss_ipc_free', self.own)
                raise RuntimeError('the ranks of the group are not on one node')
            self.mapped = []
            ptrs = []
            for r, (_, pid, hb) in enumerate(everyone):
                if r == self.rank:
                    ptrs.append(self.own)
                    continue
                if pid == os.getpid():
                    raise RuntimeError('two ranks in one process')
                p_ = ctypes.c_void_p()
                call('tss_ipc_open', ctypes.create_string_buffer(hb, 64), ctypes.byref(p_))
                self.mapped.append(p_.value)
                ptrs.append(p_.value)
        self.peers = (ctypes.c_void_p * self.world)(*ptrs)
        self.counters = torch.zeros(N.lib().tss_bn_xchg_counters(), dtype=torch.int64, device=device)
        if self.world > 1:
            dist.barrier(group=group)          # every rank has mapped every mailbox before anybody writes

    def error(self):
        out = ctypes.c_long(0)
        with torch.cuda.device(self.device):
            call('tss_bn_xchg_error', self.own, ctypes.byref(out))
        return out.value


def _exchange(group, device):
    """The IPC exchange of `group` on `device`, or None (switched off, set-up failed: the caller uses dist.all_reduce)."""
    if not syncbn_ipc:
        return None
    key = (id(group), device.index)
    if key not in _XCHG:
        try:
            _XCHG[key] = _Exchange(group, device)
        except Exception as exc:       # noqa: BLE001 -- any failure of the IPC set-up: the collective library still works
            import warnings
            warnings.warn('SyncBatchNorm: HIP IPC exchange unavailable (%s); using dist.all_reduce' % (exc,))
            _XCHG[key] = None
    return _XCHG[key]


def _allreduce_stats(slabs, count, C, group, st):
    """slab rows of this replica -> [2C+1] f64 (sums, second moments, count), summed over the ranks."""
    import torch.distributed as dist
    vec = torch.empty(2 * C + 1, dtype=torch.float64, device=slabs.device)
    call('tss_slab_reduce', ptr(slabs), float(count), ptr(vec), C, st)
    dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)
    return vec


def as_deferred(x):
    if isinstance(x, Deferred):
        return x
    _check_device(x)
    N.dtype_code(x.dtype)
    return Deferred(to_nhwc(x))


def materialize(d, relu_override=None):
    """Deferred -> ordinary tensor (applies the pending BatchNorm/ReLU with one `join` pass)."""
    if not isinstance(d, Deferred):
        return d
    if d.link is None and not d.relu:
        return d.raw
    return join(d, None, d.relu)


# ----------------------------------------------------------------------------- convolution unit

class UnitCfg:
    __slots__ = ('kind', 'stride', 'dil', 'in_link', 'in_relu', 'bn', 'training', 'out_dtype', 'out_link',
                 'image_f32', 'cin', 'cout', 'params', 'res_fork', 'stash_fork', 'overlapped', 'kh', 'kw', 'drop_p', 'prev_join')


def _classify(conv, x_is_image):
    k = conv.kernel_size
    if tuple(k) in ((1, 3), (3, 1)):
        # factorized convolution (TSS/models/lednet.py:157-180): 3 taps along one axis, padding = dilation on that axis
        ax = 0 if tuple(k) == (1, 3) else 1                 # 0: along W, 1: along H
        d = conv.dilation[1 - ax]
        want_pad, want_dil = ((0, d), (1, d)) if ax == 0 else ((d, 0), (d, 1))
        if (conv.padding_mode != 'zeros' or conv.groups != 1 or tuple(conv.stride) != (1, 1) or tuple(conv.padding) != want_pad
                or tuple(conv.dilation) != want_dil or x_is_image):
            raise NotImplementedError('HIP path: 1x3 / 3x1 convolutions must be dense, stride 1, padding = dilation on the '
                                      'kernel axis: %r' % conv)
        return 'dense1d_w' if ax == 0 else 'dense1d_h', 1, d
    kh, kw = k
    if conv.groups == 1 and not x_is_image and kh % 2 == 1 and kw % 2 == 1 and (kh != kw or kh > 3):
        # general dense convolution (csrc/convgemm.hip, kh x kw tap grid): the 5x5 / 7x7 stride-2 layers of LEDNet's APN decoder
        # (TSS/models/lednet.py:63-64), the 1x5 / 5x1 layers of ESNet's FCUBlock (TSS/models/esnet.py:83-113)
        dd = [conv.dilation[i] for i in (0, 1) if k[i] > 1]
        d = dd[0]
        if (conv.padding_mode != 'zeros' or conv.stride[0] != conv.stride[1] or any(v != d for v in dd)
                or tuple(conv.padding) != (d * (kh - 1) // 2 if kh > 1 else 0, d * (kw - 1) // 2 if kw > 1 else 0)):
            raise NotImplementedError('HIP path: dense k x k convolutions need padding = dilation * (k - 1) / 2 and one stride: %r' % conv)
        return 'ckk', conv.stride[0], d
    if conv.padding_mode != 'zeros' or k[0] != k[1] or conv.stride[0] != conv.stride[1] \
            or conv.dilation[0] != conv.dilation[1] or conv.padding[0] != conv.padding[1]:
        raise NotImplementedError('HIP path: square kernels/strides/dilations with zero padding only: %r' % conv)
    k, s, d, p = k[0], conv.stride[0], conv.dilation[0], conv.padding[0]
    if k == 1:
        if conv.groups != 1 or s != 1 or p != 0:
            raise NotImplementedError('HIP path: 1x1 convolutions must be dense, stride 1, padding 0: %r' % conv)
        return 'pw', 1, 1
    if k != 3 or p != d:
        raise NotImplementedError('HIP path: 3x3 convolutions need padding == dilation: %r' % conv)
    if conv.groups == conv.in_channels == conv.out_channels and conv.groups > 1:
        return 'dw', s, d
    if conv.groups != 1:
        raise NotImplementedError('HIP path: grouped convolutions other than depthwise are not on the hot path: %r' % conv)
    if x_is_image:
        if d != 1 or conv.in_channels * 9 > 64:
            raise NotImplementedError('HIP path: image stem needs dilation 1 and in_channels*9 <= 64: %r' % conv)
        return 'stem', s, 1
    return 'dense', s, d


_KEEP = object()
_drop_mask_probe = None       # a list while a test collects the masks drawn by dropout-on-load convolutions


def drop_conv_supported(x, conv, p):
    """nn.Dropout(p) -> 1x1 `conv` (no BatchNorm behind it) can run as ONE unit that applies the dropout on load (conv_unit's drop_p)
    instead of a pass of its own over the activation: the Classifier tails (TSS/models/fastscnn.py:96-97, contextnet.py:85-86)."""
    if not (fuse_dropout and fuse_dropout_conv) or N.fast_paths_disabled() or not (0.0 < p < 1.0):
        return False
    try:
        kind = _classify(conv, False)[0]
    except NotImplementedError:
        return False
    raw = x.raw if isinstance(x, Deferred) else x
    if kind != 'pw' or raw.dim() != 4 or raw.dtype != torch.bfloat16 or raw.shape[1] != conv.in_channels:
        return False
    P, dt = npix(raw), N.dtype_code(raw.dtype)
    return bool(N.lib().tss_pwconv_drop_supported(P, conv.in_channels, conv.out_channels, dt)
                and N.lib().tss_pwconv_bwd_fused_drop_supported(P, conv.in_channels, conv.out_channels, dt))


def conv_unit(x, conv, bn=None, relu=False, out_dtype=None, weight=None, bias=_KEEP, gamma=None, beta=None, cout=None, drop_p=0.0):
    """conv -> [BatchNorm] -> [ReLU] as ONE deferred unit.  `x` is a Deferred, an NHWC/NCHW activation tensor
    or (for the stem) the contiguous NCHW image.  Returns a Deferred.
    drop_p > 0 (after drop_conv_supported): nn.Dropout(drop_p) sits between x's pending activation and the convolution.
    weight / bias / gamma / beta / cout: run the unit with these tensors instead of the modules' own parameters (a weight padded
    with zero output rows so that a ragged channel count fills whole 8-channel vectors, a bias that a later kernel adds)."""
    is_image = (not isinstance(x, Deferred)) and x.dim() == 4 and x.shape[1] % 8 != 0
    kind, stride, dil = _classify(conv, is_image)
    cfg = UnitCfg()
    cfg.kind, cfg.stride, cfg.dil = kind, stride, dil
    cfg.drop_p = float(drop_p)
    if cfg.drop_p and (kind != 'pw' or bn is not None):
        raise RuntimeError('conv_unit: dropout on load needs a 1x1 convolution without BatchNorm (drop_conv_supported)')
    cfg.overlapped = _overlap_depth[0] > 0 and not postpone_in_overlap
    cfg.cin, cfg.cout = conv.in_channels, (cout or conv.out_channels)
    cfg.kh, cfg.kw = conv.kernel_size
    if kind == 'stem':
        _check_device(x)
        if x.dtype not in (torch.float32, torch.bfloat16):
            raise TypeError('image must be float32 or bfloat16')
        x_raw = x.contiguous()
        cfg.in_link, cfg.in_relu = None, False
        cfg.image_f32 = x_raw.dtype == torch.float32
        cfg.out_dtype = out_dtype or x_raw.dtype
        cfg.res_fork = cfg.stash_fork = cfg.prev_join = None
    else:
        d = as_deferred(x).take()
        x_raw = d.raw
        cfg.in_link, cfg.in_relu = d.link, d.relu
        cfg.image_f32 = False
        cfg.out_dtype = x_raw.dtype
        cfg.res_fork = _pending_forks.pop(id(x_raw), None) if _pending_forks else None
        cfg.stash_fork = _pending_stash.pop(id(x_raw), None) if _pending_stash else None
        cfg.prev_join = cfg.res_fork.prev_join if cfg.res_fork is not None else (_take_join(x_raw) if d.link is None and not d.relu else None)
    if x_raw.shape[1] != conv.in_channels:
        raise RuntimeError('expected %d input channels, got %d' % (conv.in_channels, x_raw.shape[1]))
    cfg.bn = bn
    cfg.training = False
    if bn is None:
        gamma = beta = None
    if bn is not None:
        if not isinstance(bn, _BatchNorm):
            raise TypeError('expected a BatchNorm module, got %r' % bn)
        cfg.training = bn.training or (bn.running_mean is None and bn.running_var is None)
        if cfg.training and bn.momentum is None:
            raise NotImplementedError('HIP path: BatchNorm with momentum=None (cumulative average) is not supported')
        if gamma is None:
            gamma, beta = bn.weight, bn.bias
    p_weight = conv.weight if weight is None else weight
    p_bias = conv.bias if bias is _KEEP else bias
    cfg.params = (p_weight, gamma, beta, p_bias)
    # the kernels read float32 parameters; a model cast with .half() / .to(torch.bfloat16) (TSS
    # scripts/contextnet/benchmark_contextnet.py:62) hands them differentiable f32 views of its 16-bit parameters (boundary
    # plumbing: the gradient flows back through the cast, and the direct-accumulation path below stays off for them)
    weight, bias = _f32(p_weight), _f32(p_bias)
    y = ConvUnitFn.apply(x_raw, weight, _f32(gamma), _f32(beta), bias, cfg)
    return Deferred(y, cfg.out_link, relu)


def conv_unit_multi(branches, conv, bn=None, relu=False):
    """conv(torch.cat(branches, dim=1)) -> [BatchNorm] -> [ReLU] for a 1x1 `conv`, with NO concatenated tensor: the kernel walks the
    contraction source by source and applies every branch's pending BatchNorm (+ the branches' common ReLU) on load
    (tss_pwconv_fwd_multi).  A branch of shape [1, C, 1, 1] is broadcast over the map (an image-pooling branch needs no upsampled
    copy).  Eval-mode / no-grad forward only; returns None when the call is outside that envelope (the caller concatenates)."""
    if torch.is_grad_enabled() or N.fast_paths_disabled() or not (2 <= len(branches) <= 6):
        return None
    if conv.kernel_size != (1, 1) or conv.groups != 1 or conv.stride != (1, 1) or conv.padding != (0, 0) or conv.in_channels != 128 * len(branches):
        return None
    if bn is not None and (not isinstance(bn, _BatchNorm) or bn.training or bn.running_mean is None):
        return None
    ds = [as_deferred(b) for b in branches]
    full = next((d.raw for d in ds if d.raw.shape[2] * d.raw.shape[3] > 1), None)
    if full is None or full.dtype != torch.bfloat16 or conv.weight.dtype != torch.float32 or conv.out_channels % 4:
        return None
    B, _, H, W = full.shape
    in_relu = ds[0].relu
    for d in ds:
        r = d.raw
        bcast = tuple(r.shape) == (1, 128, 1, 1) and B == 1
        if r.dtype != torch.bfloat16 or r.shape[1] != 128 or d.relu != in_relu or not (bcast or tuple(r.shape) == (B, 128, H, W)) or not is_nhwc(r):
            return None
    for d in ds:
        d.take()
    dev, st = full.device, stream()
    n = len(ds)
    srcs = (ctypes.c_void_p * n)(*[d.raw.data_ptr() for d in ds])
    lds = (ctypes.c_long * n)(*[0 if d.raw.shape[2] * d.raw.shape[3] == 1 else ld(d.raw) for d in ds])

    def vec(getter):
        return (ctypes.c_void_p * n)(*[(getter(d.link).data_ptr() if d.link is not None and getter(d.link) is not None else None) for d in ds])
    means, scales, biases = vec(lambda l: l.mean), vec(lambda l: l.scale), vec(lambda l: l.beta)
    Cout, P = conv.out_channels, B * H * W
    y = new_nhwc(B, Cout, H, W, torch.bfloat16, dev)
    weight = conv.weight
    call('tss_pwconv_fwd_multi', srcs, lds, means, scales, biases, n, int(in_relu), ptr(weight), _shadow(weight, 0), ptr(_f32(conv.bias)),
         ptr(y), ld(y), P, Cout, N.dtype_code(torch.bfloat16), st)
    link = None
    if bn is not None:
        gamma, beta = _f32(bn.weight), _f32(bn.bias)
        link = BNLink(Cout, P, False, gamma, beta, dev, slabs=False)
        _finalize_forward(link, bn, False, P, Cout, gamma, st)
    return Deferred(y, link, relu)


eval_epilogue = os.environ.get('TSS_EVAL_EPILOGUE', '1') != '0'   # A/B: 0 = eval-mode block outputs through the join pass, as in training


def conv_unit_joined(x, block, residual=None, relu=True):
    """Eval-mode, no-grad form of `join(block(x), residual, relu)` for block = FusedSequential(1x1 Conv2d, BatchNorm2d): the frozen
    BatchNorm, the skip and the ReLU run in the epilogue of the convolution (tss_pwconv_fwd_joined), so the block output is written
    once and no join pass exists (BottleneckBlock.forward of both models under model.eval()).  Returns the materialised tensor, or None
    when the call is outside that envelope (training statistics, gradients, f32, hooks: the caller takes the ordinary path)."""
    if not eval_epilogue or torch.is_grad_enabled() or N.fast_paths_disabled():
        return None
    mods = list(block) if isinstance(block, torch.nn.Sequential) else None
    if mods is None or len(mods) != 2 or not isinstance(mods[0], torch.nn.Conv2d) or not isinstance(mods[1], _BatchNorm):
        return None
    conv, bn = mods
    if bn.training or bn.running_mean is None or bn.running_var is None:
        return None
    try:
        kind = _classify(conv, False)[0]
    except NotImplementedError:
        return None
    raw = x.raw if isinstance(x, Deferred) else x
    if (kind != 'pw' or raw.dim() != 4 or raw.dtype != torch.bfloat16 or raw.shape[1] != conv.in_channels
            or conv.in_channels % 8 or conv.in_channels > 768 or conv.out_channels % 8 or conv.weight.dtype != torch.float32):
        return None
    B, _, H, W = raw.shape
    Cout, P = conv.out_channels, B * H * W
    if residual is not None:
        residual = to_nhwc(materialize(residual))
        if tuple(residual.shape) != (B, Cout, H, W) or residual.dtype != raw.dtype:
            return None
    d = as_deferred(x).take()
    dev, st = raw.device, stream()
    gamma, beta = _f32(bn.weight), _f32(bn.bias)
    link = BNLink(Cout, P, False, gamma, beta, dev, slabs=False)
    _finalize_forward(link, bn, False, P, Cout, gamma, st)
    y = new_nhwc(B, Cout, H, W, raw.dtype, dev)
    call('tss_pwconv_fwd_joined', ptr(d.raw), ld(d.raw), *_aff(d.link), int(d.relu), ptr(conv.weight), _shadow(conv.weight, 0),
         ptr(_f32(conv.bias)), ptr(link.mean), ptr(link.scale), ptr(beta), ptr(residual), ld(residual) if residual is not None else 0,
         int(bool(relu)), ptr(y), ld(y), P, conv.in_channels, Cout, N.dtype_code(raw.dtype), st)
    return y


eval_bottleneck = os.environ.get('TSS_BNECK_EVAL', '1') != '0'     # A/B: 0 = eval-mode inverted residuals layer by layer


def bottleneck_eval(x, conv1, conv2, conv3):
    """Eval-mode, no-grad form of a whole inverted residual -- conv1 (1x1, BatchNorm, ReLU) -> conv2 (depthwise 3x3, BatchNorm, ReLU) ->
    conv3 (1x1, BatchNorm) -> (+ x) -> ReLU, each convN a FusedSequential as the block builders make them -- as ONE kernel
    (csrc/bneck.hip: the expanded tensors live in LDS only; BottleneckBlock.forward of both models under model.eval(),
    TSS/models/fastscnn.py:152-161, TSS/models/contextnet.py:139-147).  Returns the materialised block output, or None when the call
    is outside that envelope (gradients, batch statistics, f32, hooks, channel counts: the caller goes layer by layer)."""
    if not eval_bottleneck or torch.is_grad_enabled() or N.fast_paths_disabled():
        return None
    if not (torch.is_tensor(x) and x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16):
        return None
    parts = []
    for blk, want_relu in ((conv1, True), (conv2, True), (conv3, False)):
        mods = list(blk) if isinstance(blk, torch.nn.Sequential) else None
        if mods is None or len(mods) != (3 if want_relu else 2) or not isinstance(mods[0], torch.nn.Conv2d) \
                or not isinstance(mods[1], _BatchNorm) or (want_relu and not isinstance(mods[2], torch.nn.ReLU)):
            return None
        conv, bn = mods[0], mods[1]
        if bn.training or bn.running_mean is None or bn.running_var is None or conv.bias is not None \
                or conv.weight.dtype != torch.float32 or (bn.weight is not None and bn.weight.dtype != torch.float32):
            return None
        parts.append((conv, bn))
    (c1, bn1), (c2, bn2), (c3, bn3) = parts
    try:
        k1, k2, k3 = _classify(c1, False), _classify(c2, False), _classify(c3, False)
    except NotImplementedError:
        return None
    if k1[0] != 'pw' or k3[0] != 'pw' or k2[0] != 'dw' or k2[2] != 1 or k2[1] not in (1, 2):
        return None
    B, Cin, H, W = x.shape
    Cmid, Cout, stride = c1.out_channels, c3.out_channels, k2[1]
    if c1.in_channels != Cin or c2.in_channels != Cmid or c3.in_channels != Cmid:
        return None
    residual = int(stride == 1 and Cin == Cout)
    dt = N.dtype_code(x.dtype)
    if not N.lib().tss_bneck_eval_supported(Cin, Cmid, Cout, stride, residual, dt):
        return None
    x = to_nhwc(x)
    dev, st = x.device, stream()
    links = []
    for bn, C in ((bn1, Cmid), (bn2, Cmid), (bn3, Cout)):
        gamma, beta = _f32(bn.weight), _f32(bn.bias)
        link = BNLink(C, 1, False, gamma, beta, dev, slabs=False)
        _finalize_forward(link, bn, False, 1, C, gamma, st)
        links.append((link, beta))
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = new_nhwc(B, Cout, Ho, Wo, x.dtype, dev)
    aff = []
    for link, beta in links:
        aff += [ptr(link.mean), ptr(link.scale), ptr(beta if beta is not None else torch.zeros_like(link.mean))]
    call('tss_bneck_eval_fwd', ptr(x), ld(x), ptr(c1.weight), _shadow(c1.weight, 0), aff[0], aff[1], aff[2],
         ptr(c2.weight), aff[3], aff[4], aff[5], ptr(c3.weight), _shadow(c3.weight, 0), aff[6], aff[7], aff[8],
         residual, ptr(y), ld(y), B, H, W, Cin, Cmid, Cout, stride, dt, st)
    return y


def _f32(p):
    if p is None or p.dtype == torch.float32:
        return p
    if p.dtype not in (torch.float16, torch.bfloat16):
        raise TypeError('HIP path: parameters must be float32, float16 or bfloat16, got %s' % p.dtype)
    return p.float()


class _F32Buffers:
    """float32 stand-ins for the running statistics of a BatchNorm whose buffers were cast to 16 bits by model.half():
    the finalize kernels update the stand-ins, close() writes them back."""

    def __init__(self, bn, track):
        self.pairs = []
        self.mean = self.var = None
        if track:
            self.mean, self.var = self._get(bn.running_mean), self._get(bn.running_var)

    def _get(self, buf):
        if buf.dtype == torch.float32:
            return buf
        tmp = buf.float()
        self.pairs.append((buf, tmp))
        return tmp

    def close(self):
        for buf, tmp in self.pairs:
            buf.copy_(tmp)


def _colsum(e, C, st):
    """Per-channel sums of an NHWC tensor over its pixels (f32 [C]) through the statistics slab rows: fixed order, any C."""
    P = npix(e)
    Cw = C
    if C % 8 and ld(e) >= round_up(C, 8) and e.data_ptr() % 16 == 0:
        Cw = round_up(C, 8)       # whole 8-channel vectors (the vectorised kernel): the neighbouring columns of the buffer are summed too, and dropped
    slab = torch.empty((N.stat_slabs(), 2 * Cw), dtype=torch.float64, device=e.device)
    vec = torch.empty(2 * Cw + 1, dtype=torch.float64, device=e.device)
    call('tss_tensor_stats', ptr(e), ld(e), P, Cw, ptr(slab), N.dtype_code(e.dtype), st)
    call('tss_slab_reduce', ptr(slab), float(P), ptr(vec), Cw, st)
    return vec[:C].float()


def _bias_grad_into(e, P, Cout, out, dt, st):
    """out (f32 [Cout], pre-filled) += column sums of e"""
    if Cout <= 64:
        call('tss_bias_grad', ptr(e), ld(e), P, Cout, ptr(out), dt, st)
    else:
        out.add_(_colsum(e, Cout, st))


def _finalize_forward(link, bn, training, P, Cout, gamma, st):
    """Statistics slab rows (training) or running statistics (eval) -> mean / invstd / scale of `link`; running statistics updated."""
    if training:
        track = bn.track_running_stats and bn.running_mean is not None
        rbuf = _F32Buffers(bn, track)
        run_args = (ptr(rbuf.mean), ptr(rbuf.var),
                    ptr(bn.num_batches_tracked) if (track and bn.num_batches_tracked is not None) else None)
        link.sync = _sync_group(bn)
        ex = _exchange(link.sync, link.stats.device) if (link.sync is not None and Cout <= 768) else None
        if ex is not None:
            # slab rows -> statistics of ALL ranks inside ONE kernel (mailboxes over HIP IPC): no collective, no extra launch
            call('tss_bn_finalize_xchg', ptr(link.stats), float(P), ex.peers, ex.rank, ex.world, ptr(ex.counters), ptr(gamma),
                 float(bn.eps), float(bn.momentum), *run_args, ptr(link.mean), ptr(link.invstd), ptr(link.scale), Cout, st)
        elif link.sync is not None:
            gs = _allreduce_stats(link.stats, P, Cout, link.sync, st)
            call('tss_bn_finalize_sync', ptr(gs), ptr(gamma), float(bn.eps), float(bn.momentum), *run_args,
                 ptr(link.mean), ptr(link.invstd), ptr(link.scale), Cout, st)
        else:
            call('tss_bn_finalize', ptr(link.stats), float(P), ptr(gamma), float(bn.eps),
                 float(bn.momentum), *run_args, ptr(link.mean), ptr(link.invstd), ptr(link.scale), Cout, st)
        rbuf.close()
    else:
        pre = _EVAL_AFFINES.get(id(bn)) if _EVAL_AFFINES else None
        if pre is not None:     # written by the model-wide launch at the top of this forward
            link.mean, link.invstd, link.scale = pre.unbind(0)
        else:
            rbuf = _F32Buffers(bn, True)
            call('tss_bn_eval_affine', ptr(gamma), ptr(rbuf.mean), ptr(rbuf.var),
                 float(bn.eps), ptr(link.mean), ptr(link.invstd), ptr(link.scale), Cout, st)


class ConvUnitFn(Function):
    @staticmethod
    def forward(ctx, x, weight, gamma, beta, bias, cfg):
        dev = x.device
        B = x.shape[0]
        dt = N.dtype_code(cfg.out_dtype)
        s, d = cfg.stride, cfg.dil
        Hin, Win = x.shape[2], x.shape[3]
        Ho, Wo = (Hin - 1) // s + 1, (Win - 1) // s + 1
        Cout = cfg.cout
        y = new_nhwc(B, Cout, Ho, Wo, cfg.out_dtype, dev)
        P = B * Ho * Wo
        link = None
        if cfg.bn is not None:
            if cfg.training and P <= 1:
                raise ValueError('Expected more than 1 value per channel when training, got input size %s'
                                 % (tuple(y.shape),))
            link = BNLink(Cout, P, cfg.training, gamma, beta, dev)
        stats = ptr(link.stats) if (link is not None and cfg.training) else None
        aff = _aff(cfg.in_link)
        st = stream()
        mask = None
        if cfg.kind == 'pw' and cfg.drop_p:
            # the mask is drawn by its own small launch (one byte per 8 channels), applied on load here and in the backward sweep
            mask = torch.empty((P, 16), dtype=torch.uint8, device=dev)
            counter = _dropout_counter(dev)
            call('tss_dropout_mask', ptr(counter), ptr(mask), P, cfg.cin, cfg.drop_p, st)
            if _drop_mask_probe is not None:         # tests read the mask back to hand the oracle the same one
                _drop_mask_probe.append(mask)
            call('tss_pwconv_fwd_drop', ptr(x), ld(x), *aff, int(cfg.in_relu), ptr(weight), _shadow(weight, 0), ptr(bias),
                 ptr(y), ld(y), ptr(mask), cfg.drop_p, ptr(counter), P, cfg.cin, Cout, dt, st)
        elif cfg.kind == 'pw':
            call('tss_pwconv_fwd', ptr(x), ld(x), *aff, int(cfg.in_relu), ptr(weight), _shadow(weight, 0), ptr(bias),
                 ptr(y), ld(y), stats, P, cfg.cin, Cout, dt, st)
        elif cfg.kind == 'dw':
            if bias is not None:
                raise NotImplementedError('HIP path: depthwise convolution with bias')
            call('tss_dwconv3x3_fwd', ptr(x), ld(x), *aff, int(cfg.in_relu), ptr(weight),
                 ptr(y), ld(y), stats, B, Hin, Win, Cout, s, d, dt, st)
        elif cfg.kind == 'ckk' or (cfg.kind == 'dense' and bias is not None):
            nt = cfg.kh * cfg.kw
            w_tnc = torch.empty((nt, Cout, cfg.cin), dtype=torch.float32, device=dev)
            call('tss_permute_wtaps', ptr(weight), ptr(w_tnc), None, Cout, cfg.cin, nt, st)
            call('tss_convkxk_fwd', ptr(x), ld(x), *aff, int(cfg.in_relu), ptr(w_tnc), ptr(bias), ptr(y), ld(y), stats,
                 B, Hin, Win, cfg.cin, Cout, cfg.kh, cfg.kw, s, d, dt, st)
        elif cfg.kind == 'dense':
            w_tnc = w_tnc16 = None
            if _conv3x3_stream(x.dtype, cfg.cin, Cout, s, cfg.in_link, cfg.in_relu) or _conv3x3_lean(x.dtype, cfg.cin, Cout, s, d):
                # bf16 tap-major copy: the weight-stationary / register-streamed kernels (wstat.hip, atrous.hip), the LDS-halo kernel (conv3x3.hip)
                w_tnc16 = _W3X3.get(weight.data_ptr()) if _W3X3 else None          # prepared ahead of the forward (Dense3x3Shadows)
                if w_tnc16 is None:
                    w_tnc16 = torch.empty((9, Cout, cfg.cin), dtype=torch.bfloat16, device=dev)
                    call('tss_permute_w3x3_bf16', ptr(weight), ptr(w_tnc16), None, Cout, cfg.cin, st)
            else:
                w_tnc = torch.empty((9, Cout, cfg.cin), dtype=torch.float32, device=dev)
                call('tss_permute_w3x3', ptr(weight), ptr(w_tnc), None, Cout, cfg.cin, st)
            call('tss_conv3x3_fwd', ptr(x), ld(x), *aff, int(cfg.in_relu), ptr(w_tnc), ptr(w_tnc16),
                 ptr(y), ld(y), stats, B, Hin, Win, cfg.cin, Cout, s, d, dt, st)
        elif cfg.kind in ('dense1d_w', 'dense1d_h'):
            axis = 0 if cfg.kind == 'dense1d_w' else 1
            if N.lib().tss_conv1d3_lean_supported(cfg.cin, Cout, dt):      # reads the layer's own weight tensor
                call('tss_conv1d3_fwd_w', ptr(x), ld(x), *aff, int(cfg.in_relu), ptr(weight), ptr(bias), ptr(y), ld(y), stats,
                     B, Hin, Win, cfg.cin, Cout, axis, d, dt, st)
            else:
                w_tnc = torch.empty((3, Cout, cfg.cin), dtype=torch.float32, device=dev)
                call('tss_permute_wtaps', ptr(weight), ptr(w_tnc), None, Cout, cfg.cin, 3, st)
                call('tss_conv1d3_fwd', ptr(x), ld(x), *aff, int(cfg.in_relu), ptr(w_tnc), ptr(bias), ptr(y), ld(y), stats,
                     B, Hin, Win, cfg.cin, Cout, axis, d, dt, st)
        else:  # stem
            if bias is not None:
                raise NotImplementedError('HIP path: stem convolution with bias')
            call('tss_stem3x3_fwd', ptr(x), int(cfg.image_f32), ptr(weight), ptr(y), ld(y), stats,
                 B, cfg.cin, Hin, Win, Cout, s, dt, st)
        if link is not None:
            _finalize_forward(link, cfg.bn, cfg.training, P, Cout, gamma, st)
        cfg.out_link = link
        ctx.cfg = cfg
        ctx.save_for_backward(x, weight, y if link is not None else None, mask)
        ctx.has_bias = bias is not None
        ctx.has_affine = gamma is not None
        return y

    @staticmethod
    def backward(ctx, e):
        cfg = ctx.cfg
        x, weight, y, drop_mask = ctx.saved_tensors
        dev = x.device
        e = to_nhwc(e)
        dt = N.dtype_code(e.dtype)
        st = stream()
        link = cfg.out_link
        B, Hin, Win = x.shape[0], x.shape[2], x.shape[3]
        s, d = cfg.stride, cfg.dil
        Cout, Cin = cfg.cout, cfg.cin
        P = npix(e)
        p_weight, p_gamma, p_beta, p_bias = cfg.params
        dgamma = dbeta = None
        if link is not None:
            acc = 0
            if ctx.has_affine:
                dgamma, dbeta = _direct_target(p_gamma), _direct_target(p_beta)
                if dgamma is not None and dbeta is not None:
                    acc = 1
                else:
                    dgb = torch.empty((2, Cout), dtype=torch.float32, device=dev)
                    dgamma, dbeta = dgb[0], dgb[1]
            _bn_bwd_finalize(link, Cout, acc, dgamma, dbeta, st)       # (local or cross-replica statistics: link.sync)
            if acc:
                dgamma = dbeta = None
            ga, gb, gce, gmu = link.ga, link.gb, link.gce, link.mean
            if not link.training:
                y, gb, gce, gmu = None, None, None, None
        else:
            ga = gb = gce = gmu = None
            y = None
        gargs = (ptr(e), ld(e), ptr(y), ld(y) if y is not None else 0, ptr(ga), ptr(gb), ptr(gce), ptr(gmu))
        il = cfg.in_link

        dw = _direct_target(p_weight)
        dw_ret = None
        if dw is None:
            dw = dw_ret = torch.zeros_like(weight)
        need_dx = ctx.needs_input_grad[0]
        e_in = None
        # backward-weight and backward-data of one layer are independent: the weight gradient goes to a side stream
        # (fork after the finalize above, join before this function returns, so every tensor it reads is still alive
        # and a HIP-graph capture sees two parallel branches)
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev) if (overlap_wgrad and need_dx and cfg.kind != 'stem' and drop_mask is None
                                    and overlap_min_elems <= P * (Cin + Cout) <= overlap_max_elems) else None
        wst = st
        if side is not None:
            side.wait_stream(main)
            wst = side.cuda_stream
        fused_pw = False
        fused_dbias = None
        rows_after_join = None
        if cfg.kind == 'stem':
            ws = torch.empty((N.stat_slabs(), Cout * 28), dtype=torch.float32, device=dev)
            call('tss_stem3x3_bwd_weight', *gargs, ptr(x), int(cfg.image_f32), ptr(dw), ptr(ws),
                 B, Cin, Hin, Win, Cout, s, dt, st)
            if need_dx:
                raise NotImplementedError('HIP path: gradient with respect to the input image is not implemented')
        else:
            xargs = (ptr(x), ld(x), *_aff(il), int(cfg.in_relu))
            deferred_in = il is not None or cfg.in_relu
            if cfg.kind == 'pw':
                # few channels, many pixels: input gradient and weight gradient in ONE sweep (csrc/pwbwd.hip): e, y, x read once
                fused_pw = bool(fuse_pw_backward and need_dx and side is None and e.dtype == torch.bfloat16
                                and not N.fast_paths_disabled() and N.lib().tss_pwconv_bwd_fused_preferred(P, Cin, Cout, dt))
                if drop_mask is not None:        # dropout on load: the one sweep is the only backward that knows the mask
                    fused_pw = True
            postponed = False
            sweep_pw = False
            if cfg.kind == 'pw' and not fused_pw:
                # the large layers (128 -> 128, 64 -> 384, 384 -> 64 with enough pixels): one sweep with the whole weight-gradient tile in
                # the registers of one 512-thread block per CU (csrc/pwsweep.hip)
                fork = getattr(cfg, 'res_fork', None)
                radd_s = fork.g2 if fork is not None else None
                sweep_pw = bool(sweep_pw_backward and need_dx and side is None and e.dtype == torch.bfloat16 and x.dtype == torch.bfloat16
                                and not N.fast_paths_disabled()
                                and N.lib().tss_pwconv_bwd_sweep_preferred(P, Cin, Cout, int(bool(deferred_in)), dt)
                                and (radd_s is None or (not deferred_in and radd_s.dtype == e.dtype and tuple(radd_s.shape) == tuple(x.shape)
                                                        and is_nhwc(radd_s))))
            if fused_pw or sweep_pw:
                pass
            elif cfg.kind == 'pw':
                nws = N.lib().tss_pwconv_bwd_weight_ws(P, Cin, Cout, dt) if y is not None else 0
                ws = torch.empty(nws, dtype=torch.float32, device=dev) if nws else None
                defer = 1 if (ws is not None and need_dx and side is None) else 0   # a backward-data launch carries the reduce
                # the launch itself waits for the next BatchNorm-backward finalize of this pass and carries it (see _Pass.wg)
                postponed = bool(defer and postpone_wgrad and dw_ret is None and not getattr(cfg, 'overlapped', False)
                                 and _backward_task() != -1 and not _param_observed(p_weight))
                if not postponed:
                    _flush_wg(st)
                    _flush_red(st)
                    call('tss_pwconv_bwd_weight', *gargs, *xargs, ptr(dw), ptr(ws), defer, P, Cin, Cout, dt, None, wst)
            elif cfg.kind == 'dw':
                ws = torch.empty((N.stat_slabs(), Cout * 9), dtype=torch.float32, device=dev)
                defer = 1 if (need_dx and side is None) else 0      # backward-data carries the row reduction
                # input gradient and weight gradient in ONE sweep when the shape allows (e, y, x read once)
                fused_dw = bool(defer and ((fuse_dw_backward and N.lib().tss_dwconv3x3_bwd_fused_supported(Cout, s, d, dt))
                                          or N.lib().tss_dwconv3x3_bwd_fused_preferred(Cout, s, d, dt)))
                if not fused_dw:
                    call('tss_dwconv3x3_bwd_weight', *gargs, *xargs, ptr(dw), ptr(ws), defer, B, Hin, Win, Cout, s, d, dt, wst)
            elif cfg.kind in ('dense1d_w', 'dense1d_h'):
                axis = 0 if cfg.kind == 'dense1d_w' else 1
                rows = N.lib().tss_conv1d3_bwd_weight_rows(P, Cin, Cout, dt)
                rows_w = 0 if rows else N.lib().tss_convtap_bwd_weight_rows(P, Cin, Cout, 3, dt)      # 128-channel layers (csrc/fcg.hip)
                if rows_w:
                    ws = torch.empty((rows_w, Cout * Cin * 3), dtype=torch.float32, device=dev)
                    call('tss_convtap_bwd_weight_sweep', *gargs, *xargs, ptr(ws), B, Hin, Win, Cin, Cout, 3, axis, d, dt, wst)
                    rows_after_join = (ws, rows_w, Cout * Cin * 3)
                elif rows:
                    # one sweep over e, y and x (csrc/fc1d.hip); the rows of per-block partial sums are added to the gradient together with
                    # those of every other such layer, in one launch at the end of this backward pass
                    ws = torch.empty((rows, Cout * Cin * 3), dtype=torch.float32, device=dev)
                    call('tss_conv1d3_bwd_weight_sweep', *gargs, *xargs, ptr(ws), B, Hin, Win, Cin, Cout, axis, d, dt, wst)
                    rows_after_join = (ws, rows, Cout * Cin * 3)      # the sweep may be on the side stream: its rows are queued / added after the join below
                elif (e.dtype == torch.bfloat16 and Cin % 8 == 0 and Cout % 8 == 0 and unfold_1d_wgrad and not N.fast_paths_disabled()):
                    # unfold once (bf16 [P][Cin*3], column c*3 + tap), then the pipelined pointwise MFMA weight-gradient kernel with
                    # K = 3*Cin writes torch's [N][Cin][1][3] / [N][Cin][3][1] layout directly (as the dense 3x3 below)
                    col = torch.empty((P, Cin * 3), dtype=torch.bfloat16, device=dev)
                    call('tss_im2col1d3', *xargs, ptr(col), B, Hin, Win, Cin, axis, d, dt, wst)
                    g_ = gargs
                    if y is None:        # no BatchNorm behind this layer (or a frozen one): g = ga * e, written as ga * e + 0 * e + 0
                        one, zero = _const_rows(dev, Cout)
                        g_ = (ptr(e), ld(e), ptr(e), ld(e), ptr(ga) if ga is not None else ptr(one), ptr(zero), ptr(zero), ptr(zero))
                    nws = N.lib().tss_pwconv_bwd_weight_ws(P, Cin * 3, Cout, dt)
                    ws = torch.empty(nws, dtype=torch.float32, device=dev) if nws else None
                    call('tss_pwconv_bwd_weight', *g_, ptr(col), Cin * 3, None, None, None, 0, ptr(dw), ptr(ws), 0,
                         P, Cin * 3, Cout, dt, None, wst)
                else:
                    call('tss_conv1d3_bwd_weight', *gargs, *xargs, ptr(dw), B, Hin, Win, Cin, Cout, axis, d, dt, wst)
            elif (cfg.kind == 'ckk' and s == 1 and (cfg.kh, cfg.kw) in ((1, 5), (5, 1))
                  and N.lib().tss_convtap_bwd_weight_rows(P, Cin, Cout, 5, dt)):
                # ESNet's 64-channel 1x5 / 5x1 layers: one sweep over e, y and x (csrc/fcg.hip), rows added at the end of the pass
                rows = N.lib().tss_convtap_bwd_weight_rows(P, Cin, Cout, 5, dt)
                ws = torch.empty((rows, Cout * Cin * 5), dtype=torch.float32, device=dev)
                call('tss_convtap_bwd_weight_sweep', *gargs, *xargs, ptr(ws), B, Hin, Win, Cin, Cout, 5, 0 if cfg.kh == 1 else 1, d, dt, wst)
                rows_after_join = (ws, rows, Cout * Cin * 5)
            elif cfg.kind == 'ckk':
                call('tss_convkxk_bwd_weight', *gargs, *xargs, ptr(dw), B, Hin, Win, Cin, Cout, cfg.kh, cfg.kw, s, d, dt, wst)
            elif (cfg.kind == 'dense' and s == 2 and d == 1 and N.lib().tss_sconv_bwd_weight_rows(B, Hin, Win, Cin, Cout, dt)):
                # stride-2 3x3 of the downsampling blocks: one sweep over e and x (csrc/sconv.hip), rows added at the end of the pass
                rows = N.lib().tss_sconv_bwd_weight_rows(B, Hin, Win, Cin, Cout, dt)
                ws = torch.empty((rows, Cout * Cin * 9), dtype=torch.float32, device=dev)
                call('tss_sconv_bwd_weight_sweep', *gargs, *xargs, ptr(ws), B, Hin, Win, Cin, Cout, dt, wst)
                rows_after_join = (ws, rows, Cout * Cin * 9)
            elif (e.dtype == torch.bfloat16 and s == 1 and y is not None and (Cin * 9) % 8 == 0 and Cout % 8 == 0
                  and not N.fast_paths_disabled()):
                # unfold once (bf16 [P][Cin*9], column c*9 + tap), then the pointwise MFMA weight-gradient kernel with
                # K = 9*Cin writes torch's [N][Cin][3][3] layout directly
                col = torch.empty((P, Cin * 9), dtype=torch.bfloat16, device=dev)
                call('tss_im2col3x3', *xargs, ptr(col), B, Hin, Win, Cin, d, dt, wst)
                nws = N.lib().tss_pwconv_bwd_weight_ws(P, Cin * 9, Cout, dt)
                ws = torch.empty(nws, dtype=torch.float32, device=dev) if nws else None
                call('tss_pwconv_bwd_weight', *gargs, ptr(col), Cin * 9, None, None, None, 0, ptr(dw), ptr(ws), 0,
                     P, Cin * 9, Cout, dt, None, wst)
            else:
                call('tss_conv3x3_bwd_weight', *gargs, *xargs, ptr(dw), B, Hin, Win, Cin, Cout, s, d, dt, wst)
            if need_dx or drop_mask is not None:
                e_in = new_nhwc(B, Cin, Hin, Win, e.dtype, dev)
                margs = xargs if deferred_in else (None, 0, None, None, None, 0)
                bst = ptr(il.bstats) if il is not None else None
                if fused_pw:
                    rows = (N.lib().tss_pwconv_bwd_fused_drop_rows(P) if drop_mask is not None
                            else N.lib().tss_pwconv_bwd_fused_rows(P, Cin, Cout))
                    ws = torch.empty((rows, Cout * Cin), dtype=torch.float32, device=dev)
                    bws = dbias = None
                    if ctx.has_bias:          # the bias gradient leaves the same sweep as per-block rows (no colsum launch, no atomics)
                        bws = torch.empty((rows, Cout), dtype=torch.float32, device=dev)
                        dbias = _direct_target(p_bias)
                        if dbias is None:
                            dbias = fused_dbias = torch.zeros(Cout, dtype=torch.float32, device=dev)
                    if drop_mask is not None:
                        call('tss_pwconv_bwd_fused_drop', *gargs, ptr(weight), *xargs, int(bool(deferred_in)), ptr(drop_mask),
                             cfg.drop_p, ptr(e_in), ld(e_in), bst, ptr(ws), ptr(bws), P, Cin, Cout, dt, st)
                    else:
                        call('tss_pwconv_bwd_fused', *gargs, ptr(weight), _shadow(weight, 1), *xargs, int(bool(deferred_in)),
                             ptr(e_in), ld(e_in), bst, ptr(ws), ptr(bws), P, Cin, Cout, dt, st)
                    if dw_ret is None and fused_dbias is None and batch_dw_reductions:
                        _defer_dw_reduction(ws, dw, Cout * Cin, rows, p_weight)       # summed with the depthwise rows, at the end of the pass
                        if bws is not None:
                            _defer_dw_reduction(bws, dbias, Cout, rows, p_bias)
                    else:
                        _reduce_rows_now(ws, dw, Cout * Cin, rows)
                        if bws is not None:
                            _reduce_rows_now(bws, dbias, Cout, rows)
                elif sweep_pw:
                    rows = N.lib().tss_pwconv_bwd_sweep_rows(P, Cin, Cout)
                    ws = torch.empty((rows, Cout * Cin), dtype=torch.float32, device=dev)
                    wT = _shadow(weight, 1)
                    hold_wT = None
                    if wT is None:          # no current shadow (plain autograd outside a Trainer): a transpose of this call's own
                        hold_wT = weight.detach().reshape(Cout, Cin).t().contiguous().to(torch.bfloat16)
                        wT = ptr(hold_wT)
                    call('tss_pwconv_bwd_sweep', *gargs, wT, *xargs, int(bool(deferred_in)), ptr(radd_s),
                         ld(radd_s) if radd_s is not None else 0, ptr(e_in), ld(e_in), bst, ptr(ws), P, Cin, Cout, dt, st)
                    del hold_wT
                    if radd_s is not None:
                        fork.consumed = True
                    if dw_ret is None and batch_dw_reductions:
                        _defer_dw_reduction(ws, dw, Cout * Cin, rows, p_weight)       # summed with the depthwise rows, at the end of the pass
                    else:
                        _reduce_rows_now(ws, dw, Cout * Cin, rows)
                elif cfg.kind == 'pw':
                    fork = getattr(cfg, 'res_fork', None)
                    radd = fork.g2 if fork is not None else None
                    if postponed:      # this launch carries the slot reduction of an EARLIER layer's weight gradient, if one waits
                        red, hold = _take_red(dev)
                    else:
                        red, hold = ((ptr(ws), ptr(dw), 0, 0, 0) if defer else (None, None, 0, 0, 0)), None
                    radd_ok = (radd is not None and not deferred_in and y is not None and e.dtype == torch.bfloat16 and radd.dtype == e.dtype
                               and tuple(radd.shape) == tuple(e_in.shape) and N.lib().tss_pwconv_bwd_data_radd_supported(P, Cin, Cout, dt))
                    pj = getattr(cfg, 'prev_join', None)
                    if (pj is not None and (radd is None or radd_ok) and not deferred_in and y is not None and e.dtype == torch.bfloat16
                            and P * Cin <= fuse_join_backward_max
                            and pj.j_a is not None and pj.j_ptr == x.data_ptr() and tuple(pj.j_a.shape) == tuple(x.shape)
                            and is_nhwc(pj.j_a) and N.lib().tss_pwconv_bwd_data_radd_supported(P, Cin, Cout, dt)):
                        # this layer's input is a block output: that join's backward (ReLU mask + BatchNorm-backward sums) in the epilogue
                        jl = pj.a_link
                        call('tss_pwconv_bwd_data_joined', *gargs, ptr(weight), _shadow(weight, 1), ptr(e_in), ld(e_in), *red,
                             ptr(radd), ld(radd) if radd is not None else 0, ptr(x), ld(x), ptr(pj.j_a), ld(pj.j_a), ptr(jl.mean),
                             ptr(jl.bstats), P, Cin, Cout, dt, st)
                        pj.fused_e = e_in
                        if radd is not None:
                            fork.consumed = True
                    elif radd_ok:
                        call('tss_pwconv_bwd_data_radd', *gargs, ptr(weight), _shadow(weight, 1), ptr(e_in), ld(e_in),
                             *red, ptr(radd), ld(radd), P, Cin, Cout, dt, st)
                        fork.consumed = True
                    else:
                        call('tss_pwconv_bwd_data', *gargs, ptr(weight), _shadow(weight, 1), *margs, ptr(e_in), ld(e_in), bst,
                             *red, P, Cin, Cout, dt, st)
                    del hold
                    if postponed:
                        _flush_wg(st)          # at most one weight gradient waits per stream
                        own = torch.cuda.current_stream(dev)

                        ps = _cur_pass()

                        def launch_wg(fin, _keep=(e, y, x, link, il, ws, dw), _args=(gargs, xargs, ptr(dw), ptr(ws), P, Cin, Cout, dt),
                                      _sid=st, _own=own, _ps=ps):
                            ga_, xa_, dwp, wsp, P_, K_, N_, dt_ = _args
                            call('tss_pwconv_bwd_weight', *ga_, *xa_, dwp, wsp, 1, P_, K_, N_, dt_,
                                 ctypes.byref(fin) if fin is not None else None, _sid)
                            _flush_red(_sid, _ps)
                            with _lock:
                                _ps.red[_sid] = (_keep[5], _keep[6], P_, K_, N_, _keep[0].device, _own)
                        with _lock:
                            ps.wg[st] = (launch_wg, dev, own)
                elif cfg.kind == 'dw' and fused_dw and dw_ret is None and batch_dw_reductions:
                    # the rows of per-block partial sums stay in ws; they are added to the (direct) gradient together with
                    # those of every other depthwise layer, in one launch at the end of this backward pass
                    rows = ctypes.c_int(0)
                    call('tss_dwconv3x3_bwd_fused_sweep', *gargs, ptr(weight), *xargs, int(bool(deferred_in)), ptr(e_in), ld(e_in),
                         bst, ptr(ws), B, Hin, Win, Cout, s, d, dt, st, ctypes.byref(rows))
                    _defer_dw_reduction(ws, dw, Cout * 9, rows.value, p_weight)
                elif cfg.kind == 'dw' and fused_dw:
                    call('tss_dwconv3x3_bwd_fused', *gargs, ptr(weight), *xargs, int(bool(deferred_in)), ptr(e_in), ld(e_in),
                         bst, ptr(ws), ptr(dw), B, Hin, Win, Cout, s, d, dt, st)
                elif cfg.kind == 'dw':
                    call('tss_dwconv3x3_bwd_data', *gargs, ptr(weight), *margs, ptr(e_in), ld(e_in), bst,
                         ptr(ws) if defer else None, ptr(dw) if defer else None, B, Hin, Win, Cout, s, d, dt, st)
                elif cfg.kind in ('dense1d_w', 'dense1d_h'):
                    if N.lib().tss_conv1d3_lean_supported(Cin, Cout, dt):
                        call('tss_conv1d3_bwd_data_w', *gargs, ptr(weight), *margs, ptr(e_in), ld(e_in), bst,
                             B, Hin, Win, Cin, Cout, 0 if cfg.kind == 'dense1d_w' else 1, d, dt, st)
                    else:
                        w_tcn = torch.empty((3, Cin, Cout), dtype=torch.float32, device=dev)
                        call('tss_permute_wtaps', ptr(weight), None, ptr(w_tcn), Cout, Cin, 3, st)
                        call('tss_conv1d3_bwd_data', *gargs, ptr(w_tcn), *margs, ptr(e_in), ld(e_in), bst,
                             B, Hin, Win, Cin, Cout, 0 if cfg.kind == 'dense1d_w' else 1, d, dt, st)
                elif cfg.kind == 'ckk' or s != 1:
                    # general tap grid / transposed gather of a strided layer (generic implicit-GEMM kernel)
                    nt = cfg.kh * cfg.kw
                    w_tcn = torch.empty((nt, Cin, Cout), dtype=torch.float32, device=dev)
                    call('tss_permute_wtaps', ptr(weight), None, ptr(w_tcn), Cout, Cin, nt, st)
                    call('tss_convkxk_bwd_data', *gargs, ptr(w_tcn), *margs, ptr(e_in), ld(e_in), bst,
                         B, Hin, Win, Cin, Cout, cfg.kh, cfg.kw, s, d, dt, st)
                else:
                    w_tcn = w_tcn16 = None
                    if _conv3x3_lean(e.dtype, Cout, Cin, 1, d):       # contraction over Cout, outputs = Cin
                        w_tcn16 = torch.empty((9, Cin, Cout), dtype=torch.bfloat16, device=dev)
                        call('tss_permute_w3x3_bf16', ptr(weight), None, ptr(w_tcn16), Cout, Cin, st)
                    else:
                        w_tcn = torch.empty((9, Cin, Cout), dtype=torch.float32, device=dev)
                        call('tss_permute_w3x3', ptr(weight), None, ptr(w_tcn), Cout, Cin, st)
                    call('tss_conv3x3_bwd_data', *gargs, ptr(w_tcn), ptr(w_tcn16), *margs, ptr(e_in), ld(e_in), bst,
                         B, Hin, Win, Cin, Cout, d, dt, st)
        stash = getattr(cfg, 'stash_fork', None)
        if stash is not None and e_in is not None and not deferred_in:
            stash.g2 = e_in                # the other consumer of this layer's input adds it in its own backward (fork_two)
        dbias_ret = fused_dbias
        if ctx.has_bias and not (fused_pw and (need_dx or drop_mask is not None)):
            dbias = _direct_target(p_bias)
            if link is not None and link.training:
                # BatchNorm with batch statistics removes any per-channel shift: d(loss)/d(bias) is exactly zero
                if dbias is None:
                    dbias_ret = torch.zeros(Cout, dtype=torch.float32, device=dev)
            elif link is not None:          # frozen statistics: d(bias) = scale * sum(e)
                tmp = torch.zeros(Cout, dtype=torch.float32, device=dev)
                _bias_grad_into(e, P, Cout, tmp, dt, st)
                if dbias is None:
                    dbias_ret = tmp * link.ga
                else:
                    dbias.addcmul_(tmp, link.ga)
            else:
                if dbias is None:
                    dbias = dbias_ret = torch.zeros(Cout, dtype=torch.float32, device=dev)
                _bias_grad_into(e, P, Cout, dbias, dt, st)
        if side is not None:
            main.wait_stream(side)
        if rows_after_join is not None:
            ws_, rows_, ncols_ = rows_after_join
            if dw_ret is None and batch_dw_reductions:
                _defer_dw_reduction(ws_, dw, ncols_, rows_, p_weight)
            else:
                _reduce_rows_now(ws_, dw, ncols_, rows_)
        return e_in, dw_ret, dgamma, dbeta, dbias_ret, None


# ----------------------------------------------------------------------------- join (materialise / add / relu)

class JoinCfg:
    __slots__ = ('a_link', 'b_link', 'relu', 'links', 'relus', 'drop_p', 'res_fork', 'j_a', 'j_ptr', 'fused_e', '__weakref__')


def join(a, b=None, relu=False, dropout_p=0.0, fork=None):
    """relu?(bn_a(a) + bn_b(b)) -> ordinary NHWC tensor.  a, b: Deferred (without pending ReLU) or tensors.
    dropout_p > 0 (ReLU joins only): nn.Dropout applied in the same pass, forward and backward."""
    a = as_deferred(a).take()
    cfg = JoinCfg()
    cfg.relu = bool(relu)
    cfg.drop_p = float(dropout_p)
    if cfg.drop_p and not (relu and 0.0 < cfg.drop_p < 1.0):
        raise RuntimeError('join: dropout can only be folded into a ReLU join, 0 < p < 1')
    if a.relu and (b is not None or not relu):
        raise RuntimeError('join: a pending ReLU can only be folded into a single-input relu join')
    cfg.a_link = a.link
    braw = None
    cfg.b_link = None
    if b is not None:
        b = as_deferred(b).take()
        if b.relu:
            raise RuntimeError('join: operand b carries a pending ReLU; materialize it first')
        if b.raw.shape != a.raw.shape or b.raw.dtype != a.raw.dtype:
            raise RuntimeError('join: operands differ in shape/dtype: %s vs %s' % (a.raw.shape, b.raw.shape))
        braw, cfg.b_link = b.raw, b.link
    cfg.res_fork = fork if b is not None else None
    cfg.j_a = cfg.j_ptr = cfg.fused_e = None
    out = JoinFn.apply(a.raw, braw, cfg)
    if (fuse_join_backward and cfg.relu and not cfg.drop_p and cfg.a_link is not None and cfg.b_link is None and out.requires_grad
            and out.dtype == torch.bfloat16 and not N.fast_paths_disabled()):
        import weakref
        cfg.j_a, cfg.j_ptr = a.raw, out.data_ptr()
        if len(_join_ctx) > 256:          # outputs nobody picked up (consumed by something other than a conv unit)
            _join_ctx.clear()
        _join_ctx[id(out)] = weakref.ref(cfg)
    return out


class JoinFn(Function):
    @staticmethod
    def forward(ctx, a, b, cfg):
        out = new_nhwc(*a.shape, a.dtype, a.device)
        al, bl = cfg.a_link, cfg.b_link
        slot = None
        if cfg.drop_p:
            slot = torch.empty(1, dtype=torch.int64, device=a.device)
            call('tss_dropout_tick', ptr(_dropout_counter(a.device)), ptr(slot), stream())
        call('tss_join_fwd', ptr(a), ld(a), *_aff(al), ptr(b), ld(b) if b is not None else 0, *_aff(bl),
             ptr(out), ld(out), int(cfg.relu), cfg.drop_p, ptr(slot), npix(a), a.shape[1], N.dtype_code(a.dtype), stream())
        ctx.cfg = cfg
        ctx.has_b = b is not None
        ctx.save_for_backward(a if al is not None else None, b if bl is not None else None,
                              out if cfg.relu else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        cfg = ctx.cfg
        a, b, out = ctx.saved_tensors
        fe = getattr(cfg, 'fused_e', None)
        cfg.fused_e = None
        if (fe is not None and dout.data_ptr() == fe.data_ptr() and dout.shape == fe.shape and dout.dtype == fe.dtype
                and dout.stride() == fe.stride()):
            # the consumer's backward-data launch already did this join's backward on the complete gradient (mask + slab rows written):
            # dout IS e.  (Any other gradient of `out` would have made autograd hand us a different, summed tensor.)
            fork = getattr(cfg, 'res_fork', None)
            if fork is not None and ctx.has_b:
                fork.g2 = dout
            return dout, (dout if ctx.has_b else None), None
        del fe
        dout = to_nhwc(dout)
        al, bl = cfg.a_link, cfg.b_link
        e = new_nhwc(*dout.shape, dout.dtype, dout.device) if cfg.relu else None
        if cfg.relu or al is not None or bl is not None:
            call('tss_join_bwd', ptr(dout), ld(dout), ptr(out), ld(out) if out is not None else 0, int(cfg.relu),
                 ptr(a), ld(a) if a is not None else 0, ptr(al.mean) if al else None, ptr(al.bstats) if al else None,
                 ptr(b), ld(b) if b is not None else 0, ptr(bl.mean) if bl else None, ptr(bl.bstats) if bl else None,
                 ptr(e), ld(e) if e is not None else 0, 1.0 / (1.0 - cfg.drop_p) if cfg.drop_p else 1.0,
                 npix(dout), dout.shape[1], N.dtype_code(dout.dtype), stream())
        g = e if e is not None else dout
        fork = getattr(cfg, 'res_fork', None)
        if fork is not None and ctx.has_b:
            fork.g2 = g                    # the skip's gradient, for the epilogue of the block's first layer
        return g, (g if ctx.has_b else None), None


# ----------------------------------------------------------------------------- dropout

_dropout_counters = {}


def _dropout_counter(device):
    key = (device.type, device.index)
    if key not in _dropout_counters:
        seed = torch.initial_seed() & 0x7FFFFFFFFFFFFFFF
        _dropout_counters[key] = torch.tensor([seed], dtype=torch.int64, device=device)
    return _dropout_counters[key]


def dropout(x, p, training):
    """nn.Dropout on a materialised NHWC tensor (Philox mask, regenerated in backward)."""
    if not training or p == 0.0:
        return x
    if p >= 1.0:
        raise NotImplementedError('HIP path: dropout with p >= 1')
    return DropoutFn.apply(to_nhwc(x), float(p))


class DropoutFn(Function):
    @staticmethod
    def forward(ctx, x, p):
        slot = torch.empty(1, dtype=torch.int64, device=x.device)
        call('tss_dropout_tick', ptr(_dropout_counter(x.device)), ptr(slot), stream())
        y = new_nhwc(*x.shape, x.dtype, x.device)
        call('tss_dropout', ptr(x), ld(x), ptr(y), ld(y), npix(x), x.shape[1], p, ptr(slot),
             N.dtype_code(x.dtype), stream())
        ctx.p = p
        ctx.save_for_backward(slot)
        return y

    @staticmethod
    def backward(ctx, dy):
        (slot,) = ctx.saved_tensors
        dy = to_nhwc(dy)
        dx = new_nhwc(*dy.shape, dy.dtype, dy.device)
        call('tss_dropout', ptr(dy), ld(dy), ptr(dx), ld(dx), npix(dy), dy.shape[1], ctx.p, ptr(slot),
             N.dtype_code(dy.dtype), stream())
        return dx, None


# ----------------------------------------------------------------------------- resampling

def _out_size(x, size, scale_factor):
    if size is not None:
        return (size, size) if isinstance(size, int) else (int(size[0]), int(size[1]))
    sf = (scale_factor, scale_factor) if not isinstance(scale_factor, (tuple, list)) else scale_factor
    # torch: output = floor(input * scale_factor)
    import math
    return int(math.floor(x.shape[2] * sf[0])), int(math.floor(x.shape[3] * sf[1]))


def bilinear(x, size=None, scale_factor=None, out=None):
    """F.interpolate(mode='bilinear', align_corners=True) on an NHWC activation tensor."""
    x = to_nhwc(materialize(x))
    ho, wo = _out_size(x, size, scale_factor)
    return BilinearFn.apply(x, ho, wo)


class BilinearFn(Function):
    @staticmethod
    def forward(ctx, x, ho, wo):
        B, C, H, W = x.shape
        y = new_nhwc(B, C, ho, wo, x.dtype, x.device)
        call('tss_bilinear_nhwc_fwd', ptr(x), ld(x), ptr(y), ld(y), B, H, W, ho, wo, C,
             N.dtype_code(x.dtype), stream())
        ctx.geom = (B, C, H, W, ho, wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W, ho, wo = ctx.geom
        dy = to_nhwc(dy)
        dx = new_nhwc(B, C, H, W, dy.dtype, dy.device)
        tmp = torch.empty((B * H * wo * C,), dtype=torch.float32, device=dy.device)
        call('tss_bilinear_nhwc_bwd', ptr(dy), ld(dy), ptr(dx), ld(dx), ptr(tmp), B, H, W, ho, wo, C,
             N.dtype_code(dy.dtype), stream())
        return dx, None, None


class _UpDwCfg:
    __slots__ = ('size', 'dil', 'bn', 'training', 'out_link', 'params')


def upsample_dw_unit(x, size, block):
    """Bilinear upsample (align_corners=True) to `size` followed by `block` = depthwise 3x3 conv (padding = dilation) -> [BatchNorm]
    -> [ReLU] as ONE deferred unit (csrc/updw.hip): the upsampled tensor is never written -- the low-resolution branch of the
    feature-fusion modules (TSS/models/fastscnn.py:74-76, TSS/models/contextnet.py:110-122).  Returns a Deferred, or None when
    the call is outside the kernel's envelope (the caller then runs the two operators one after the other).  TSS_UPDW=0: never."""
    mods = list(block) if isinstance(block, torch.nn.Sequential) else None
    if mods is None or not mods or N.fast_paths_disabled():
        return None
    conv = mods[0]
    bn = mods[1] if len(mods) > 1 and isinstance(mods[1], _BatchNorm) else None
    relu = isinstance(mods[-1], torch.nn.ReLU) and len(mods) > 1
    if len(mods) != 1 + int(bn is not None) + int(relu) or not isinstance(conv, torch.nn.Conv2d) or conv.bias is not None:
        return None
    try:
        kind, s, d = _classify(conv, False)
    except NotImplementedError:
        return None
    if kind != 'dw' or s != 1:
        return None
    x = to_nhwc(materialize(x))
    ho, wo = int(size[0]), int(size[1])
    B, C, Hs, Ws = x.shape
    if (x.dtype != torch.bfloat16 or C != conv.in_channels
            or not N.lib().tss_updw_supported(B, Hs, Ws, ho, wo, C, d, N.dtype_code(x.dtype))):
        return None
    cfg = _UpDwCfg()
    cfg.size, cfg.dil, cfg.bn, cfg.training = (ho, wo), d, bn, False
    gamma = beta = None
    if bn is not None:
        cfg.training = bn.training or (bn.running_mean is None and bn.running_var is None)
        if cfg.training and bn.momentum is None:
            return None
        gamma, beta = bn.weight, bn.bias
    cfg.params = (conv.weight, gamma, beta)
    y = UpDwFn.apply(x, _f32(conv.weight), _f32(gamma), _f32(beta), cfg)
    return Deferred(y, cfg.out_link, relu)


class UpDwFn(Function):
    @staticmethod
    def forward(ctx, x, weight, gamma, beta, cfg):
        dev, st = x.device, stream()
        B, C, Hs, Ws = x.shape
        ho, wo = cfg.size
        y = new_nhwc(B, C, ho, wo, x.dtype, dev)
        P = B * ho * wo
        link = None
        if cfg.bn is not None:
            if cfg.training and P <= 1:
                raise ValueError('Expected more than 1 value per channel when training, got input size %s' % (tuple(y.shape),))
            link = BNLink(C, P, cfg.training, gamma, beta, dev)
        stats = ptr(link.stats) if (link is not None and cfg.training) else None
        call('tss_updw_fwd', ptr(x), ld(x), Hs, Ws, ptr(weight), ptr(y), ld(y), stats, B, ho, wo, C, cfg.dil,
             N.dtype_code(x.dtype), st)
        if link is not None:
            _finalize_forward(link, cfg.bn, cfg.training, P, C, gamma, st)
        cfg.out_link = link
        ctx.cfg = cfg
        ctx.save_for_backward(x, weight, y if link is not None else None)
        ctx.has_affine = gamma is not None
        return y

    @staticmethod
    def backward(ctx, e):
        cfg = ctx.cfg
        x, weight, y = ctx.saved_tensors
        dev, st = x.device, stream()
        e = to_nhwc(e)
        dt = N.dtype_code(e.dtype)
        B, C, Hs, Ws = x.shape
        ho, wo = cfg.size
        link = cfg.out_link
        p_weight, p_gamma, p_beta = cfg.params
        dgamma = dbeta = None
        ga = gb = gce = gmu = None
        if link is not None:
            acc = 0
            if ctx.has_affine:
                dgamma, dbeta = _direct_target(p_gamma), _direct_target(p_beta)
                if dgamma is not None and dbeta is not None:
                    acc = 1
                else:
                    dgb = torch.empty((2, C), dtype=torch.float32, device=dev)
                    dgamma, dbeta = dgb[0], dgb[1]
            _bn_bwd_finalize(link, C, acc, dgamma, dbeta, st)
            if acc:
                dgamma = dbeta = None
            ga, gb, gce, gmu = link.ga, link.gb, link.gce, link.mean
            if not link.training:
                y, gb, gce, gmu = None, None, None, None
        else:
            y = None
        dw = _direct_target(p_weight)
        dw_ret = None
        if dw is None:
            dw = dw_ret = torch.zeros_like(weight)
        ws = torch.empty((N.lib().tss_updw_ws_rows(B, Hs, Ws, ho, wo, C, cfg.dil, dt), C * 9), dtype=torch.float32, device=dev)
        e_up = new_nhwc(B, C, ho, wo, e.dtype, dev)
        rows = ctypes.c_int(0)
        call('tss_updw_bwd', ptr(e), ld(e), ptr(y), ld(y) if y is not None else 0, ptr(ga), ptr(gb), ptr(gce), ptr(gmu),
             ptr(weight), ptr(x), ld(x), Hs, Ws, ptr(e_up), ld(e_up), ptr(ws), B, ho, wo, C, cfg.dil, dt, st, ctypes.byref(rows))
        if dw_ret is None and batch_dw_reductions:
            _defer_dw_reduction(ws, dw, C * 9, rows.value, p_weight)
        else:
            _reduce_rows_now(ws, dw, C * 9, rows.value)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = new_nhwc(B, C, Hs, Ws, e.dtype, dev)
            tmp = torch.empty((B * Hs * wo * C,), dtype=torch.float32, device=dev)
            call('tss_bilinear_nhwc_bwd', ptr(e_up), ld(e_up), ptr(dx), ld(dx), ptr(tmp), B, Hs, Ws, ho, wo, C, dt, st)
        return dx, dw_ret, dgamma, dbeta, None


def resize_image(x, size=None, scale_factor=None, out_dtype=None):
    """Bilinear (align_corners=True) resize of the NCHW image itself (ContextNet's context branch input,
    TSS/models/contextnet.py:65-67).  No gradient flows to the image."""
    _check_device(x)
    x = x.contiguous()
    ho, wo = _out_size(x, size, scale_factor)
    B, C, H, W = x.shape
    out_dtype = out_dtype or x.dtype
    y = torch.empty((B, C, ho, wo), dtype=out_dtype, device=x.device)
    call('tss_bilinear_planar_fwd', ptr(x), N.dtype_code(x.dtype), ptr(y), N.dtype_code(out_dtype),
         B * C, H, W, ho, wo, stream())
    return y


def upsample_logits(low, scale_factor=None, size=None):
    """The decoder head's final F.interpolate: NHWC low-res logits -> NCHW-contiguous full-res logits."""
    low = to_nhwc(materialize(low))
    ho, wo = _out_size(low, size, scale_factor)
    if wo % 8:
        raise NotImplementedError('HIP path: logits width must be a multiple of 8, got %d' % wo)
    return UpsampleHeadFn.apply(low, ho, wo)


class Upsample(torch.nn.Module):
    """nn.Upsample(scale_factor=k, mode='bilinear', align_corners=True) on the HIP path: the auxiliary heads of the
    reference's deep-supervision recipe (scripts/train_fastscnn.py:108-117) end in one.  NHWC logits in, NCHW-contiguous
    full-resolution logits out (what the losses consume), through tss_upsample_head_{fwd,bwd}."""

    def __init__(self, size=None, scale_factor=None, mode='bilinear', align_corners=True):
        super().__init__()
        if mode != 'bilinear' or not align_corners:
            raise NotImplementedError("HIP path: Upsample(mode='bilinear', align_corners=True) only")
        self.size, self.scale_factor = size, scale_factor

    def forward(self, input):
        return upsample_logits(input, scale_factor=self.scale_factor, size=self.size)

    def extra_repr(self):
        return 'size=%r, scale_factor=%r, mode=bilinear, align_corners=True' % (self.size, self.scale_factor)


class UpsampleHeadFn(Function):
    @staticmethod
    def forward(ctx, low, ho, wo):
        B, C, h, w = low.shape
        y = torch.empty((B, C, ho, wo), dtype=low.dtype, device=low.device)
        call('tss_upsample_head_fwd', ptr(low), ld(low), ptr(y), B, C, h, w, ho, wo,
             N.dtype_code(low.dtype), stream())
        ctx.geom = (B, C, h, w, ho, wo, ld(low))
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, h, w, ho, wo, ldl = ctx.geom
        dy = dy.contiguous()
        tmp = torch.empty((B * C * h * wo,), dtype=torch.float32, device=dy.device)
        # padded channels of dlow must be finite zeros: they are read (and discarded) by 8-wide loaders
        base = torch.zeros((B, h, w, ldl), dtype=dy.dtype, device=dy.device)
        dlow = base.permute(0, 3, 1, 2)[:, :C]
        call('tss_upsample_head_bwd', ptr(dy), None, ptr(tmp), ptr(dlow), ldl, B, C, h, w, ho, wo,
             N.dtype_code(dy.dtype), stream())
        return dlow, None, None


def adaptive_avg_pool(x, bins):
    """nn.AdaptiveAvgPool2d(bins).  Few, large windows (global average pooling: the image-pooling branch of an ASPP head pools ONE
    256 x 512 window per image at 2048 x 4096 -- 1.05 ms through the one-block-per-window kernel) go through the row-sliced
    pyramid-pooling kernel; many small windows through the one-block-per-window kernel."""
    x = to_nhwc(materialize(x))
    bins = int(bins)
    if x.shape[0] * bins * bins <= 128 and x.shape[2] * x.shape[3] >= 4096:
        return PoolMultiFn.apply(x, (bins,), None)[0]
    return AdaptivePoolFn.apply(x, bins)


class AdaptivePoolFn(Function):
    @staticmethod
    def forward(ctx, x, bins):
        B, C, H, W = x.shape
        y = new_nhwc(B, C, bins, bins, x.dtype, x.device)
        call('tss_adaptive_pool_fwd', ptr(x), ld(x), ptr(y), ld(y), B, H, W, C, bins,
             N.dtype_code(x.dtype), stream())
        ctx.geom = (B, C, H, W, bins)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W, bins = ctx.geom
        dy = to_nhwc(dy)
        dx = new_nhwc(B, C, H, W, dy.dtype, dy.device)
        call('tss_adaptive_pool_bwd', ptr(dy), ld(dy), ptr(dx), ld(dx), B, H, W, C, bins,
             N.dtype_code(dy.dtype), stream())
        return dx, None


def concat_upsampled(x, branches):
    """torch.cat([x, *[bilinear(b, size=x.shape[2:]) for b in branches]], dim=1) written in place into one
    NHWC buffer (TSS/models/fastscnn.py:118-122)."""
    x = to_nhwc(materialize(x))
    branches = [to_nhwc(materialize(b)) for b in branches]
    return ConcatUpFn.apply(x, *branches)


class ConcatUpFn(Function):
    @staticmethod
    def forward(ctx, x, *branches):
        B, C, H, W = x.shape
        ctot = C + sum(b.shape[1] for b in branches)
        out = new_nhwc(B, ctot, H, W, x.dtype, x.device)
        dt, st = N.dtype_code(x.dtype), stream()
        call('tss_copy_nhwc', ptr(x), ld(x), ptr(out), ld(out), npix(x), C, dt, st)
        off = C
        geoms = []
        for b in branches:
            cb, hb, wb = b.shape[1], b.shape[2], b.shape[3]
            sl = out[:, off:off + cb]
            call('tss_bilinear_nhwc_fwd', ptr(b), ld(b), ptr(sl), ld(out), B, hb, wb, H, W, cb, dt, st)
            geoms.append((off, cb, hb, wb))
            off += cb
        ctx.geom = (B, C, H, W, geoms)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, C, H, W, geoms = ctx.geom
        dout = to_nhwc(dout)
        dt, st = N.dtype_code(dout.dtype), stream()
        grads = [dout[:, :C]]
        for off, cb, hb, wb in geoms:
            db = new_nhwc(B, cb, hb, wb, dout.dtype, dout.device)
            sl = dout[:, off:off + cb]
            tmp = torch.empty((B * hb * W * cb,), dtype=torch.float32, device=dout.device)
            call('tss_bilinear_nhwc_bwd', ptr(sl), ld(dout), ptr(db), ld(db), ptr(tmp), B, hb, wb, H, W, cb, dt, st)
            grads.append(db)
        return tuple(grads)


def channel_shuffle(x, groups):
    """channel_shuffle of TSS/models/lednet.py:183-188 on an NHWC activation (the channel index is the fastest axis, so this
    is a permutation inside every pixel row)."""
    x = to_nhwc(materialize(x))
    if x.shape[1] % groups or x.shape[1] % 8:
        raise RuntimeError('channel_shuffle: channels must be a multiple of groups and of 8')
    return ChannelShuffleFn.apply(x, int(groups))


class ChannelShuffleFn(Function):
    @staticmethod
    def forward(ctx, x, groups):
        y = new_nhwc(*x.shape, x.dtype, x.device)
        call('tss_channel_shuffle', ptr(x), ld(x), ptr(y), ld(y), npix(x), x.shape[1], groups, N.dtype_code(x.dtype), stream())
        ctx.groups = groups
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = to_nhwc(dy)
        dx = new_nhwc(*dy.shape, dy.dtype, dy.device)
        call('tss_channel_shuffle', ptr(dy), ld(dy), ptr(dx), ld(dx), npix(dy), dy.shape[1], dy.shape[1] // ctx.groups,
             N.dtype_code(dy.dtype), stream())
        return dx, None


def channel_slice(x, C):
    """x[:, :C] of an NHWC tensor computed with padded channels: a view; its gradient is padded back in one pass (tss_pad_channels)."""
    if C == x.shape[1]:
        return x
    return ChannelSliceFn.apply(x, int(C))


class ChannelSliceFn(Function):
    @staticmethod
    def forward(ctx, x, C):
        ctx.cp = x.shape[1]
        return x[:, :C]

    @staticmethod
    def backward(ctx, g):
        B, C, H, W = g.shape
        pitch = g.stride(3)
        if not (g.is_cuda and g.stride(1) == 1 and pitch >= C and g.stride(2) == W * pitch and g.stride(0) == H * W * pitch):
            g = to_nhwc(g)          # (any pitch is fine for the kernel; only a non-channels-last gradient is re-laid out)
        out = new_nhwc(B, ctx.cp, H, W, g.dtype, g.device)
        call('tss_pad_channels', ptr(g), g.stride(3), C, ptr(out), ld(out), ctx.cp, B * H * W, N.dtype_code(g.dtype), stream())
        return out, None


def split_fork(x):
    """(x[:, :C/2], x[:, C/2:], x) for a unit that runs two branches on torch.chunk(input, 2, 1) and adds the whole input back
    (SSnbtBlock, TSS/models/lednet.py:112-124): three views, no copy; the gradient is cat(g_left, g_right) + g_skip in ONE pass
    (tss_cat2_add) instead of autograd's two zero-filled tensors, two slice copies and two additions."""
    x = to_nhwc(materialize(x))
    if x.shape[1] % 16:
        raise NotImplementedError('HIP path: a split unit needs a multiple of 16 channels')
    return SplitForkFn.apply(x)


class SplitForkFn(Function):
    @staticmethod
    def forward(ctx, x):
        half = x.shape[1] // 2
        return x[:, :half], x[:, half:], x.view_as(x)

    @staticmethod
    def backward(ctx, gl, gr, gs):
        ref = gl if gl is not None else (gr if gr is not None else gs)
        if ref is None:
            return None
        B, _, H, W = ref.shape
        half = (ref.shape[1] // 2) if (gl is None and gr is None) else ref.shape[1]
        if gl is None and gr is None:
            return gs
        gl = to_nhwc(gl) if gl is not None else new_nhwc(B, half, H, W, ref.dtype, ref.device).zero_()
        gr = to_nhwc(gr) if gr is not None else new_nhwc(B, half, H, W, ref.dtype, ref.device).zero_()
        gs = to_nhwc(gs) if gs is not None else None
        out = new_nhwc(B, 2 * half, H, W, ref.dtype, ref.device)
        call('tss_cat2_add', ptr(gl), ld(gl), ptr(gr), ld(gr), ptr(gs), ld(gs) if gs is not None else 0, ptr(out), ld(out),
             B * H * W, half, N.dtype_code(ref.dtype), stream())
        return out


fuse_ssnbt_tail = os.environ.get('TSS_SSNBT_TAIL', '1') != '0'     # A/B: 0 = concat_joined, channel_dropout, join, channel_shuffle as four operators


def ssnbt_tail(left, right, x, drop_p, training):
    """channel_shuffle(relu(x + dropout2d(cat([bn(left), bn(right)], 1))), 2): the tail of SSnbtBlock.forward (TSS/models/lednet.py:112-124)
    in one pass each way (csrc/ssnbt.hip).  left / right: Deferreds whose BatchNorm (no ReLU) is still pending; x: the unit's input."""
    dl, dr = as_deferred(left), as_deferred(right)
    x = to_nhwc(materialize(x))
    ok = (fuse_ssnbt_tail and dl.link is not None and dr.link is not None and not dl.relu and not dr.relu
          and dl.raw.shape == dr.raw.shape and dl.raw.dtype == dr.raw.dtype == x.dtype and dl.raw.shape[1] % 8 == 0
          and x.shape[1] == 2 * dl.raw.shape[1] and tuple(x.shape[2:]) == tuple(dl.raw.shape[2:]) and x.shape[0] == dl.raw.shape[0])
    if not ok:
        y = concat_joined([dl, dr], relu=False)
        y = channel_dropout(y, drop_p, training)
        return channel_shuffle(join(y, x, relu=True), 2)
    dl.take(); dr.take()
    m = None
    if training and drop_p > 0.0:
        keep = 1.0 - float(drop_p)
        m = (torch.rand((x.shape[0], x.shape[1]), device=x.device) < keep).to(torch.float32)
        if keep > 0.0:
            m = m / keep
    cfg = JoinCfg()
    cfg.links = [dl.link, dr.link]
    return SSnbtTailFn.apply(cfg, dl.raw, dr.raw, x, m)


class SSnbtTailFn(Function):
    @staticmethod
    def forward(ctx, cfg, l, r, x, m):
        B, C, H, W = x.shape
        out = new_nhwc(B, C, H, W, x.dtype, x.device)
        call('tss_ssnbt_tail_fwd', ptr(l), ld(l), *_aff(cfg.links[0]), ptr(r), ld(r), *_aff(cfg.links[1]), ptr(x), ld(x), ptr(m),
             ptr(out), ld(out), B, H * W, C, N.dtype_code(x.dtype), stream())
        ctx.cfg = cfg
        ctx.save_for_backward(out, l, r, m)
        return out

    @staticmethod
    def backward(ctx, dout):
        out, l, r, m = ctx.saved_tensors
        ll, lr = ctx.cfg.links
        dout = to_nhwc(dout)
        B, C, H, W = out.shape
        half = C // 2
        gs = new_nhwc(B, C, H, W, dout.dtype, dout.device)
        e = new_nhwc(B, C, H, W, dout.dtype, dout.device) if m is not None else None
        call('tss_ssnbt_tail_bwd', ptr(dout), ld(dout), ptr(out), ld(out), ptr(l), ld(l), ptr(ll.mean), ptr(r), ld(r), ptr(lr.mean),
             ptr(m), ptr(e), ld(e) if e is not None else 0, ptr(gs), ld(gs), ptr(ll.bstats), ptr(lr.bstats), B, H * W, C,
             N.dtype_code(dout.dtype), stream())
        ee = e if e is not None else gs
        return None, ee[:, :half], ee[:, half:], gs, None


def concat(tensors):
    """torch.cat(tensors, dim=1) of equally sized activation tensors into one NHWC buffer (tss_copy_nhwc per operand; the
    gradient is a channel slice of the incoming one, no copy).  Channel counts must be multiples of 8."""
    ts = [to_nhwc(materialize(t)) for t in tensors]
    if any(t.shape[1] % 8 or t.shape[0] != ts[0].shape[0] or t.shape[2:] != ts[0].shape[2:] or t.dtype != ts[0].dtype for t in ts):
        raise RuntimeError('concat: operands must share batch / spatial size / dtype and have channel counts that are multiples of 8')
    return ConcatFn.apply(*ts)


class ConcatFn(Function):
    @staticmethod
    def forward(ctx, *ts):
        B, _, H, W = ts[0].shape
        out = new_nhwc(B, sum(t.shape[1] for t in ts), H, W, ts[0].dtype, ts[0].device)
        dt, st, off = N.dtype_code(out.dtype), stream(), 0
        for t in ts:
            c = t.shape[1]
            call('tss_copy_nhwc', ptr(t), ld(t), ptr(out[:, off:off + c]), ld(out), npix(t), c, dt, st)
            off += c
        ctx.widths = [t.shape[1] for t in ts]
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = to_nhwc(dout)
        grads, off = [], 0
        for c in ctx.widths:
            grads.append(dout[:, off:off + c])
            off += c
        return tuple(grads)


def concat_joined(branches, relu=True):
    """torch.cat([relu?(bn_i(branch_i)) for i], dim=1): every branch is a Deferred whose pending BatchNorm(+ReLU) is applied
    while it is written into its channel slice of the result -- one pass per branch, no normalised copy, no concat copy (the
    parallel branches of an ASPP module).  Backward hands each branch its masked gradient and BatchNorm-backward sums."""
    ds = [as_deferred(b).take() for b in branches]
    if any(d.raw.shape[1] % 8 or d.raw.shape[0] != ds[0].raw.shape[0] or d.raw.shape[2:] != ds[0].raw.shape[2:]
           or d.raw.dtype != ds[0].raw.dtype for d in ds):
        raise RuntimeError('concat_joined: branches must share batch / spatial size / dtype, channels multiples of 8')
    cfg = JoinCfg()
    cfg.links = [d.link for d in ds]
    cfg.relus = [bool(relu) or bool(d.relu) for d in ds]
    return ConcatJoinFn.apply(cfg, *[d.raw for d in ds])


class ConcatJoinFn(Function):
    @staticmethod
    def forward(ctx, cfg, *raws):
        B, _, H, W = raws[0].shape
        out = new_nhwc(B, sum(r.shape[1] for r in raws), H, W, raws[0].dtype, raws[0].device)
        dt, st, off = N.dtype_code(out.dtype), stream(), 0
        for r, link, relu in zip(raws, cfg.links, cfg.relus):
            c = r.shape[1]
            call('tss_join_fwd', ptr(r), ld(r), *_aff(link), None, 0, None, None, None, ptr(out[:, off:off + c]), ld(out),
                 int(relu), 0.0, None, npix(r), c, dt, st)
            off += c
        ctx.cfg = cfg
        ctx.save_for_backward(out, *raws)
        return out

    @staticmethod
    def backward(ctx, dout):
        out, *raws = ctx.saved_tensors
        cfg = ctx.cfg
        dout = to_nhwc(dout)
        dt, st, off, grads = N.dtype_code(dout.dtype), stream(), 0, []
        for r, link, relu in zip(raws, cfg.links, cfg.relus):
            c = r.shape[1]
            dsl, osl = dout[:, off:off + c], out[:, off:off + c]
            if relu or link is not None:
                e = new_nhwc(*r.shape, dout.dtype, dout.device) if relu else None
                call('tss_join_bwd', ptr(dsl), ld(dout), ptr(osl), ld(out), int(relu), ptr(r) if link is not None else None, ld(r),
                     ptr(link.mean) if link is not None else None, ptr(link.bstats) if link is not None else None,
                     None, 0, None, None, ptr(e), ld(e) if e is not None else 0, 1.0, npix(r), c, dt, st)
                grads.append(e if e is not None else dsl)
            else:
                grads.append(dsl)
            off += c
        return (None, *grads)


# ----------------------------------------------------------------------------- channel-attention gate (BiSeNet)

def gate(x, a, add_one=False):
    """x * (sigmoid(a) + add_one) with a = one value per image and channel ([B, C, 1, 1], the 1x1 convolution's output on the
    pooled map before the sigmoid): TSS/models/bisenet.py:128-131 (add_one) and :144-148."""
    x = to_nhwc(materialize(x))
    a = to_nhwc(materialize(a))
    if a.shape != (x.shape[0], x.shape[1], 1, 1) or a.dtype != x.dtype:
        raise RuntimeError('gate: attention must be (B, C, 1, 1) of the map\'s dtype, got %s %s for %s %s'
                           % (tuple(a.shape), a.dtype, tuple(x.shape), x.dtype))
    if x.shape[1] % 8:
        raise NotImplementedError('HIP path: gate needs a channel count that is a multiple of 8')
    return GateFn.apply(x, a, 1.0 if add_one else 0.0)


class GateFn(Function):
    @staticmethod
    def forward(ctx, x, a, add_one):
        B, C, H, W = x.shape
        out = new_nhwc(B, C, H, W, x.dtype, x.device)
        call('tss_gate_fwd', ptr(x), ld(x), ptr(a), ld(a), ptr(out), ld(out), B, H * W, C, add_one, N.dtype_code(x.dtype), stream())
        ctx.add_one = add_one
        ctx.save_for_backward(x, a)
        return out

    @staticmethod
    def backward(ctx, g):
        x, a = ctx.saved_tensors
        B, C, H, W = x.shape
        g = to_nhwc(g)
        dx = new_nhwc(B, C, H, W, x.dtype, x.device)
        da = new_nhwc(B, C, 1, 1, x.dtype, x.device)
        ws = torch.empty(B * N.lib().tss_gate_slices(B, H * W) * C, dtype=torch.float32, device=x.device)
        call('tss_gate_bwd', ptr(g), ld(g), ptr(x), ld(x), ptr(a), ld(a), ptr(dx), ld(dx), ptr(da), ld(da), ptr(ws), B, H * W, C,
             ctx.add_one, N.dtype_code(x.dtype), stream())
        return dx, da, None


# ----------------------------------------------------------------------------- pyramid pooling, all arms per launch

fuse_dropout = True    # nn.Dropout after a pending BatchNorm + ReLU rides in the join that materialises it (False: own pass)
# nn.Dropout in front of a 1x1 convolution is applied on load by that convolution (no join at all; TSS_DROP_CONV=0: the join form)
fuse_dropout_conv = os.environ.get('TSS_DROP_CONV', '1') != '0'
ppm_fused = os.environ.get('TSS_PPM_FUSED', '1') != '0'   # False: every arm through the generic operators (A/B checks)


# ----------------------------------------------------------------------------- LEDNet / ESNet glue (csrc/zoo.hip)

class _BNCfg:
    __slots__ = ('bn', 'training', 'link', 'params')


def batch_norm(z, bn, relu=False, gamma=None, beta=None):
    """BatchNorm2d (+ReLU) on a MATERIALISED tensor as a deferred unit: the statistics are computed now (one pass over z), the
    normalisation is applied by the consumer's load like after a convolution.  The nn.BatchNorm2d that follows
    torch.cat([conv(x), pool(x)]) in DownsamplingBlock (TSS/models/lednet.py:126-144, TSS/models/esnet.py:47-68)."""
    z = to_nhwc(materialize(z))
    if not isinstance(bn, _BatchNorm):
        raise TypeError('expected a BatchNorm module, got %r' % bn)
    if z.shape[1] % 8 or z.shape[1] != bn.num_features:
        raise NotImplementedError('HIP path: batch_norm needs %d channels, a multiple of 8; got %d' % (bn.num_features, z.shape[1]))
    cfg = _BNCfg()
    cfg.bn = bn
    cfg.training = bn.training or (bn.running_mean is None and bn.running_var is None)
    if cfg.training and bn.momentum is None:
        raise NotImplementedError('HIP path: BatchNorm with momentum=None (cumulative average) is not supported')
    if gamma is None:
        gamma, beta = bn.weight, bn.bias
    cfg.params = (gamma, beta)
    y = StandaloneBNFn.apply(z, _f32(gamma), _f32(beta), cfg)
    return Deferred(y, cfg.link, relu)


class StandaloneBNFn(Function):
    @staticmethod
    def forward(ctx, z, gamma, beta, cfg):
        C, P = z.shape[1], npix(z)
        if cfg.training and P <= 1:
            raise ValueError('Expected more than 1 value per channel when training, got input size %s' % (tuple(z.shape),))
        st = stream()
        link = BNLink(C, P, cfg.training, gamma, beta, z.device)
        if cfg.training:
            call('tss_tensor_stats', ptr(z), ld(z), P, C, ptr(link.stats), N.dtype_code(z.dtype), st)
        _finalize_forward(link, cfg.bn, cfg.training, P, C, gamma, st)
        cfg.link = link
        ctx.cfg = cfg
        ctx.has_affine = gamma is not None
        ctx.save_for_backward(z)
        return z.view_as(z)

    @staticmethod
    def backward(ctx, e):
        cfg = ctx.cfg
        z, = ctx.saved_tensors
        link = cfg.link
        e = to_nhwc(e)
        C, P = z.shape[1], npix(z)
        st = stream()
        dgamma = dbeta = None
        acc = 0
        if ctx.has_affine:
            dgamma, dbeta = _direct_target(cfg.params[0]), _direct_target(cfg.params[1])
            if dgamma is not None and dbeta is not None:
                acc = 1
            else:
                dgb = torch.empty((2, C), dtype=torch.float32, device=z.device)
                dgamma, dbeta = dgb[0], dgb[1]
        _bn_bwd_finalize(link, C, acc, dgamma, dbeta, st)
        if acc:
            dgamma = dbeta = None
        dz = new_nhwc(*z.shape, e.dtype, z.device)
        tr = link.training
        call('tss_bn_bwd_apply', ptr(e), ld(e), ptr(z) if tr else None, ld(z), ptr(link.ga), ptr(link.gb) if tr else None,
             ptr(link.gce) if tr else None, ptr(link.mean) if tr else None, ptr(dz), ld(dz), P, C, N.dtype_code(e.dtype), st)
        return dz, dgamma, dbeta, None


def pool_concat(y1, bias, x, n1=None):
    """torch.cat([y1 + bias, F.max_pool2d(x, 2)], dim=1) in one pass (DownsamplingBlock, TSS/models/lednet.py:138-141 /
    TSS/models/esnet.py:62-65).  y1: the strided convolution's raw output (first n1 channels used; n1 < y1.shape[1] when the layer
    ran with zero-padded output rows), x: the block input -- an NHWC activation, or the NCHW image."""
    y1 = to_nhwc(materialize(y1))
    n1 = y1.shape[1] if n1 is None else n1
    is_image = x.dim() == 4 and x.shape[1] % 8 != 0
    if not is_image:
        x = to_nhwc(materialize(x))
    else:
        _check_device(x)
    B, Cin, Hin, Win = x.shape
    if Hin % 2 or Win % 2:
        raise NotImplementedError('HIP path: DownsamplingBlock needs even height and width (torch.cat of the reference fails otherwise)')
    if tuple(y1.shape) != (B, y1.shape[1], Hin // 2, Win // 2) or n1 > y1.shape[1]:
        raise RuntimeError('pool_concat: convolution output %s does not match the pooled input %s' % (tuple(y1.shape), tuple(x.shape)))
    if x.dtype != y1.dtype and x.dtype != torch.float32:
        raise TypeError('pool_concat: input dtype %s with activations of %s' % (x.dtype, y1.dtype))
    return PoolConcatFn.apply(y1, _f32(bias), x, n1)


class PoolConcatFn(Function):
    @staticmethod
    def forward(ctx, y1, bias, x, n1):
        B, Cin, Hin, Win = x.shape
        z = new_nhwc(B, n1 + Cin, Hin // 2, Win // 2, y1.dtype, y1.device)
        x_f32 = int(x.dtype == torch.float32)
        call('tss_pool_concat_fwd', ptr(y1), ld(y1), ptr(bias), n1, ptr(x), x_f32, *x.stride(), Cin, ptr(z), ld(z), B, Hin, Win,
             N.dtype_code(z.dtype), stream())
        ctx.n1, ctx.full1, ctx.has_bias = n1, y1.shape[1], bias is not None
        ctx.save_for_backward(x)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, = ctx.saved_tensors
        dz = to_nhwc(dz)
        B, Cin, Hin, Win = x.shape
        n1, st, dt = ctx.n1, stream(), N.dtype_code(dz.dtype)
        dev = dz.device
        if ctx.full1 == n1:
            dy1 = dz[:, :n1]
        elif ctx.full1 == dz.shape[1]:
            dy1 = dz       # the layer ran with zero-padded output rows: their gradient is discarded by the padding's own backward
        else:
            dy1 = new_nhwc(B, ctx.full1, Hin // 2, Win // 2, dz.dtype, dev)
            dy1.zero_()
            dy1[:, :n1].copy_(dz[:, :n1])
        dbias = None
        if ctx.has_bias and ctx.needs_input_grad[1]:     # column sums through the slab rows: fixed order, no atomics
            dbias = _colsum(dz, n1, st)
        dx = None
        if ctx.needs_input_grad[2]:
            dx = new_nhwc(B, Cin, Hin, Win, dz.dtype, dev)
            call('tss_pool_concat_bwd', ptr(dz), ld(dz), n1, ptr(x), int(x.dtype == torch.float32), *x.stride(), Cin,
                 ptr(dx), ld(dx), B, Hin, Win, dt, st)
        return dy1, dbias, dx, None


def conv_transpose(x, weight, bias, stride, cout=None):
    """nn.ConvTranspose2d(k odd, stride, padding = (k - 1) / 2, output_padding = stride - 1, bias) on a materialised NHWC tensor
    (UpsamplingBlock, TSS/models/esnet.py:71-80): the transposed gather of the generic implicit-GEMM kernel.  `weight` is the layer's
    [Cin][Cout][kh][kw] tensor (Cout may be zero-padded by the caller to a multiple of 8)."""
    x = to_nhwc(materialize(x))
    if weight.shape[0] != x.shape[1] or x.shape[1] % 8 or weight.shape[1] % 8 or weight.shape[2] % 2 == 0 or weight.shape[3] % 2 == 0:
        raise NotImplementedError('HIP path: transposed convolution needs channel counts that are multiples of 8 and odd kernel sides; '
                                  'got weight %s for input %s' % (tuple(weight.shape), tuple(x.shape)))
    return ConvTransposeFn.apply(x, _f32(weight), _f32(bias), int(stride))


class ConvTransposeFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride):
        B, Ci, H, W = x.shape
        Co, kh, kw = weight.shape[1], weight.shape[2], weight.shape[3]
        dev, st, dt = x.device, stream(), N.dtype_code(x.dtype)
        y = new_nhwc(B, Co, H * stride, W * stride, x.dtype, dev)
        w_tcn = torch.empty((kh * kw, Co, Ci), dtype=torch.float32, device=dev)
        call('tss_permute_wtaps', ptr(weight), None, ptr(w_tcn), Ci, Co, kh * kw, st)
        call('tss_convkxk_transposed_fwd', ptr(x), ld(x), ptr(w_tcn), ptr(bias), ptr(y), ld(y), B, H * stride, W * stride, Co, Ci,
             kh, kw, stride, dt, st)
        ctx.stride = stride
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = to_nhwc(g)
        B, Ci, H, W = x.shape
        Co, kh, kw = weight.shape[1], weight.shape[2], weight.shape[3]
        s = ctx.stride
        dev, st, dt = x.device, stream(), N.dtype_code(g.dtype)
        dx = dw = dbias = None
        if ctx.needs_input_grad[0]:        # the strided convolution of the gradient with the same weights
            w_tnc = torch.empty((kh * kw, Ci, Co), dtype=torch.float32, device=dev)
            call('tss_permute_wtaps', ptr(weight), ptr(w_tnc), None, Ci, Co, kh * kw, st)
            dx = new_nhwc(B, Ci, H, W, g.dtype, dev)
            call('tss_convkxk_fwd', ptr(g), ld(g), None, None, None, 0, ptr(w_tnc), None, ptr(dx), ld(dx), None,
                 B, H * s, W * s, Co, Ci, kh, kw, s, 1, dt, st)
        if ctx.needs_input_grad[1]:        # that convolution's weight gradient with the roles of activation and gradient swapped
            dw = torch.zeros_like(weight)
            rows = N.lib().tss_sconv_bwd_weight_rows(B, H * s, W * s, Co, Ci, dt) if (kh == 3 and kw == 3 and s == 2) else 0
            if rows:      # one sweep over x and g (csrc/sconv.hip, rectangular form), then the row reduction
                ws = torch.empty((rows, Ci * Co * 9), dtype=torch.float32, device=dev)
                call('tss_sconv_bwd_weight_sweep', ptr(x), ld(x), None, 0, None, None, None, None, ptr(g), ld(g), None, None, None, 0,
                     ptr(ws), B, H * s, W * s, Co, Ci, dt, st)
                _reduce_rows_now(ws, dw, Ci * Co * 9, rows)
            else:
                call('tss_convkxk_bwd_weight', ptr(x), ld(x), None, 0, None, None, None, None, ptr(g), ld(g), None, None, None, 0,
                     ptr(dw), B, H * s, W * s, Co, Ci, kh, kw, s, 1, dt, st)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            dbias = _colsum(g, Co, st)
        return dx, dw, dbias, None


def mul_addrows(u, a, r):
    """u * a + r with r one row per image ([B, C, 1, 1]): `x * level4(input) + level5(pooled)` of APNModule, TSS/models/lednet.py:86-90."""
    u, a, r = to_nhwc(materialize(u)), to_nhwc(materialize(a)), to_nhwc(materialize(r))
    if u.shape != a.shape or u.dtype != a.dtype or r.dtype != u.dtype or tuple(r.shape) != (u.shape[0], u.shape[1], 1, 1) or u.shape[1] % 8:
        raise RuntimeError('mul_addrows: operands %s, %s, %s' % (tuple(u.shape), tuple(a.shape), tuple(r.shape)))
    return MulAddRowsFn.apply(u, a, r)


class MulAddRowsFn(Function):
    @staticmethod
    def forward(ctx, u, a, r):
        B, C, H, W = u.shape
        out = new_nhwc(B, C, H, W, u.dtype, u.device)
        call('tss_mul_addrows_fwd', ptr(u), ld(u), ptr(a), ld(a), ptr(r), ld(r), ptr(out), ld(out), B, H * W, C,
             N.dtype_code(u.dtype), stream())
        ctx.save_for_backward(u, a)
        return out

    @staticmethod
    def backward(ctx, g):
        u, a = ctx.saved_tensors
        g = to_nhwc(g)
        B, C, H, W = u.shape
        dev = u.device
        du, da = new_nhwc(B, C, H, W, g.dtype, dev), new_nhwc(B, C, H, W, g.dtype, dev)
        dr = new_nhwc(B, C, 1, 1, g.dtype, dev)
        ws = torch.empty(B * N.lib().tss_rows_slices(B, H * W) * C, dtype=torch.float32, device=dev)
        call('tss_mul_addrows_bwd', ptr(g), ld(g), ptr(u), ld(u), ptr(a), ld(a), ptr(du), ld(du), ptr(da), ld(da), ptr(dr), ld(dr),
             ptr(ws), B, H * W, C, N.dtype_code(g.dtype), stream())
        return du, da, dr


def channel_dropout(x, p, training):
    """nn.Dropout2d (TSS/models/lednet.py:113, TSS/models/esnet.py:117,163): whole channels of an image are zeroed with probability
    p, the others scaled by 1 / (1 - p).  The mask is drawn with torch's generator (one [B, C] draw), applied by tss_scale_rows."""
    if not training or p <= 0.0:
        return x
    x = to_nhwc(materialize(x))
    if x.shape[1] % 8:
        raise NotImplementedError('HIP path: channel dropout needs a multiple of 8 channels')
    keep = 1.0 - float(p)
    m = (torch.rand((x.shape[0], x.shape[1]), device=x.device) < keep).to(torch.float32)
    if keep > 0.0:
        m = m / keep
    return ScaleRowsFn.apply(x, m)


class ScaleRowsFn(Function):
    @staticmethod
    def _run(x, m):
        B, C, H, W = x.shape
        out = new_nhwc(B, C, H, W, x.dtype, x.device)
        call('tss_scale_rows', ptr(x), ld(x), ptr(m), ptr(out), ld(out), B, H * W, C, N.dtype_code(x.dtype), stream())
        return out

    @staticmethod
    def forward(ctx, x, m):
        ctx.save_for_backward(m)
        return ScaleRowsFn._run(x, m)

    @staticmethod
    def backward(ctx, g):
        m, = ctx.saved_tensors
        return ScaleRowsFn._run(to_nhwc(g), m), None


def _hp(tensors):
    """Host array of device pointers (None -> NULL) for the tss_ppm_* entry points."""
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def _hl(vals):
    import ctypes
    return (ctypes.c_long * len(vals))(*[int(v) for v in vals])


def _hi(vals):
    import ctypes
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def adaptive_avg_pool_multi(x, bins):
    """[AdaptiveAvgPool2d(b)(x) for b in bins] from one launch (and one launch for the summed gradient)."""
    x = to_nhwc(materialize(x))
    fork = _pending_forks.pop(id(x), None) if _pending_forks else None      # x has another consumer (fork_two): its gradient is added in our backward
    return PoolMultiFn.apply(x, tuple(int(b) for b in bins), fork)


class PoolMultiFn(Function):
    @staticmethod
    def forward(ctx, x, bins, fork=None):
        ctx.fork = fork
        B, C, H, W = x.shape
        ys = [new_nhwc(B, C, b, b, x.dtype, x.device) for b in bins]
        ncells = sum(b * b for b in bins)
        S = N.lib().tss_ppm_pool_slices(B, ncells)
        ws = torch.empty(B * ncells * S * C, dtype=torch.float32, device=x.device) if S > 1 else None
        call('tss_ppm_pool_fwd', ptr(x), ld(x), _hp(ys), _hl([ld(y) for y in ys]), _hi(bins), len(bins), ptr(ws), B, H, W, C,
             N.dtype_code(x.dtype), stream())
        ctx.geom = (B, C, H, W, bins, x.dtype, x.device)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        B, C, H, W, bins, dtype, dev = ctx.geom
        dys = [to_nhwc(d) if d is not None else new_nhwc(B, C, b, b, dtype, dev).zero_() for d, b in zip(dys, bins)]
        dx = new_nhwc(B, C, H, W, dtype, dev)
        fork = ctx.fork
        radd = fork.g2 if fork is not None else None
        if radd is not None and not (radd.dtype == dtype and tuple(radd.shape) == (B, C, H, W) and is_nhwc(radd)):
            radd = None
        call('tss_ppm_pool_bwd', _hp(dys), _hl([ld(d) for d in dys]), _hi(bins), len(bins), ptr(dx), ld(dx),
             ptr(radd), ld(radd) if radd is not None else 0, B, H, W, C, N.dtype_code(dtype), stream())
        if radd is not None:
            fork.consumed = True
        return dx, None, None


def ppm_arms_fusable(x, arms):
    """concat_upsampled_arms covers these arms: <= 4 Deferred square maps of equal channel count, one slab row per cell."""
    if not ppm_fused or not 1 <= len(arms) <= 4 or not all(isinstance(a, Deferred) for a in arms):
        return False
    ca = arms[0].raw.shape[1]
    return all(a.raw.shape[1] == ca and a.raw.shape[2] == a.raw.shape[3] and a.raw.dtype == x.dtype
               and x.shape[0] * a.raw.shape[2] ** 2 <= N.stat_slabs() for a in arms) and ca % 8 == 0 and 256 % (ca // 8) == 0


def concat_upsampled_arms(x, arms):
    """torch.cat([x, *[upsample(relu(bn(arm))) for arm in arms]], 1): the pending BatchNorm + ReLU of every arm is applied
    per bilinear tap, all arms in one launch; backward hands every arm its masked gradient and BatchNorm-backward sums."""
    x = to_nhwc(materialize(x))
    ds = [a.take() for a in arms]
    cfg = JoinCfg()
    cfg.res_fork = _pending_stash.pop(id(x), None) if _pending_stash else None
    cfg.links = [d.link for d in ds]
    cfg.relus = [bool(d.relu) for d in ds]
    return PpmConcatFn.apply(x, cfg, *[d.raw for d in ds])


class PpmConcatFn(Function):
    @staticmethod
    def _tables(cfg):
        links = cfg.links
        return (_hp([l.mean if l is not None else None for l in links]), _hp([l.scale if l is not None else None for l in links]),
                _hp([l.beta if l is not None else None for l in links]), _hi(cfg.relus))

    @staticmethod
    def forward(ctx, x, cfg, *raws):
        B, C, H, W = x.shape
        n, ca = len(raws), raws[0].shape[1]
        bins = [r.shape[2] for r in raws]
        out = new_nhwc(B, C + n * ca, H, W, x.dtype, x.device)
        call('tss_ppm_concat_fwd', ptr(x), ld(x), _hp(raws), _hl([ld(r) for r in raws]), _hi(bins), *PpmConcatFn._tables(cfg),
             n, ptr(out), ld(out), B, H, W, C, ca, N.dtype_code(x.dtype), stream())
        ctx.cfg, ctx.geom = cfg, (B, C, H, W, ca, bins)
        ctx.save_for_backward(*raws)
        return out

    @staticmethod
    def backward(ctx, dout):
        raws = ctx.saved_tensors
        B, C, H, W, ca, bins = ctx.geom
        dout = to_nhwc(dout)
        es = [new_nhwc(B, ca, b, b, dout.dtype, dout.device) for b in bins]
        call('tss_ppm_concat_bwd', ptr(dout), ld(dout), _hp(raws), _hl([ld(r) for r in raws]), _hi(bins),
             *PpmConcatFn._tables(ctx.cfg), _hp([l.bstats if l is not None else None for l in ctx.cfg.links]),   # (None: sums taken on chip by ppm_arms)
             _hp(es), _hl([ld(e) for e in es]), len(raws), B, H, W, C, ca, N.dtype_code(dout.dtype), stream())
        stash = getattr(ctx.cfg, 'res_fork', None)
        if stash is not None:
            stash.g2 = dout[:, :C]         # x's other consumer (the pools) adds it in tss_ppm_pool_bwd (fork_two)
        return (dout[:, :C], None, *es)


# the arms themselves (1x1 convolution + BatchNorm statistics + finalize per arm): one launch for all of them, forward and backward
ppm_arms_fused = os.environ.get('TSS_PPM_ARMS', '1') != '0'    # '0': every arm through conv_unit (A/B checks)


def ppm_arms(blocks, pooled):
    """[block_i(pooled_i)] for the arms of a pyramid pooling module, each block = (Conv2d 1x1, BatchNorm2d[, ReLU]) -> list of
    Deferred (raw conv output + pending BatchNorm + ReLU), from ONE launch (csrc/ppm.hip); None when the arms are outside that
    kernel's envelope (the caller then runs them one by one)."""
    if not ppm_arms_fused or N.fast_paths_disabled() or not 1 <= len(blocks) <= 4:
        return None
    convs, bns, relus = [], [], []
    for blk in blocks:
        mods = list(blk)
        if len(mods) not in (2, 3) or not isinstance(mods[0], torch.nn.Conv2d) or not isinstance(mods[1], _BatchNorm):
            return None
        if len(mods) == 3 and not isinstance(mods[2], torch.nn.ReLU):
            return None
        c, bn = mods[0], mods[1]
        if (c.kernel_size != (1, 1) or c.groups != 1 or c.bias is not None or c.stride != (1, 1) or c.padding != (0, 0)
                or c.weight.dtype != torch.float32 or bn.weight is None or bn.weight.dtype != torch.float32
                or (bn.training and (bn.momentum is None or not bn.track_running_stats)) or _sync_group(bn) is not None
                or (bn.running_mean is not None and bn.running_mean.dtype != torch.float32)
                or getattr(blk, 'act_dtype', None) not in (None, torch.bfloat16)):
            return None
        convs.append(c); bns.append(bn); relus.append(len(mods) == 3)
    ps = [p.raw if isinstance(p, Deferred) and p.link is None and not p.relu else p for p in pooled]
    if any(not torch.is_tensor(p) or p.dtype != torch.bfloat16 or not is_nhwc(p) for p in ps):
        return None
    C, Ca = convs[0].in_channels, convs[0].out_channels
    training = bns[0].training
    if any(c.in_channels != C or c.out_channels != Ca for c in convs) or any(p.shape[1] != C for p in ps) \
            or any(bn.training != training or bn.eps != bns[0].eps or bn.momentum != bns[0].momentum for bn in bns):
        return None
    counts = [npix(p) for p in ps]
    if training and min(counts) <= 1:
        return None                     # conv_unit raises torch's "Expected more than 1 value per channel"
    if not N.lib().tss_ppm_arms_supported(len(ps), C, Ca, _hi(counts), N.TSS_BF16):
        return None
    cfg = JoinCfg()
    cfg.links = [None] * len(ps)
    cfg.relus = (convs, bns)
    ys = PpmArmsFn.apply(cfg, *ps, *[c.weight for c in convs], *[bn.weight for bn in bns], *[bn.bias for bn in bns])
    return [Deferred(y, link, relu) for y, link, relu in zip(ys, cfg.links, relus)]


class PpmArmsFn(Function):
    @staticmethod
    def forward(ctx, cfg, *ts):
        convs, bns = cfg.relus
        n = len(convs)
        ps, ws, gammas, betas = ts[:n], ts[n:2 * n], ts[2 * n:3 * n], ts[3 * n:4 * n]
        dev = ps[0].device
        C, Ca = convs[0].in_channels, convs[0].out_channels
        training = bns[0].training
        ys = [new_nhwc(p.shape[0], Ca, p.shape[2], p.shape[3], p.dtype, dev) for p in ps]
        counts = [npix(p) for p in ps]
        links = [BNLink(Ca, cnt, training, g_, b_, dev, slabs=False) for cnt, g_, b_ in zip(counts, gammas, betas)]
        run = [(bn.running_mean, bn.running_var, bn.num_batches_tracked) if training else (None, None, None) for bn in bns]
        call('tss_ppm_arms_fwd', _hp(ps), _hl([ld(p) for p in ps]), _hp(ws), _hp(gammas), _hp([r[0] for r in run]),
             _hp([r[1] for r in run]), _hp([r[2] for r in run]), _hp(ys), _hl([ld(y) for y in ys]), _hp([l.vec for l in links]),
             _hi(counts), n, C, Ca, int(training), float(bns[0].eps), float(bns[0].momentum if bns[0].momentum is not None else 0.0),
             N.TSS_BF16, stream())
        if not training:
            for bn, link in zip(bns, links):
                pre = _EVAL_AFFINES.get(id(bn)) if _EVAL_AFFINES else None
                if pre is not None:
                    link.mean, link.invstd, link.scale = pre.unbind(0)
                else:
                    call('tss_bn_eval_affine', ptr(bn.weight), ptr(bn.running_mean), ptr(bn.running_var), float(bn.eps),
                         ptr(link.mean), ptr(link.invstd), ptr(link.scale), Ca, stream())
        cfg.links = links
        ctx.cfg, ctx.n, ctx.geom = cfg, n, (C, Ca, counts, training)
        ctx.save_for_backward(*ps, *ws, *ys)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *es):
        n = ctx.n
        saved = ctx.saved_tensors
        ps, ws, ys = saved[:n], saved[n:2 * n], saved[2 * n:3 * n]
        convs, bns = ctx.cfg.relus
        links = ctx.cfg.links
        C, Ca, counts, training = ctx.geom
        dev = ps[0].device
        es = [to_nhwc(e) if e is not None else new_nhwc(*y.shape, y.dtype, dev).zero_() for e, y in zip(es, ys)]
        targets = [(_direct_target(c.weight), _direct_target(bn.weight), _direct_target(bn.bias)) for c, bn in zip(convs, bns)]
        direct = all(t is not None for tr in targets for t in tr)
        if not direct:
            targets = [(torch.empty_like(c.weight), torch.empty_like(bn.weight), torch.empty_like(bn.bias)) for c, bn in zip(convs, bns)]
        eins = [new_nhwc(p.shape[0], C, p.shape[2], p.shape[3], p.dtype, dev) for p in ps]
        call('tss_ppm_arms_bwd', _hp(es), _hl([ld(e) for e in es]), _hp(ys), _hl([ld(y) for y in ys]), _hp(ps), _hl([ld(p) for p in ps]),
             _hp(ws), _hp([l.gamma for l in links]), _hp([l.vec for l in links]), _hp([t[0] for t in targets]),
             _hp([t[1] for t in targets]), _hp([t[2] for t in targets]), int(direct), _hp(eins), _hl([ld(t) for t in eins]),
             _hi(counts), n, C, Ca, int(training), N.TSS_BF16, stream())
        if direct:
            return (None, *eins, *([None] * (3 * n)))
        return (None, *eins, *[t[0] for t in targets], *[t[1] for t in targets], *[t[2] for t in targets])


# ----------------------------------------------------------------------------- loss / metrics (caller side)

class CrossEntropyFn(Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        _check_device(logits)
        logits = logits.contiguous()
        B, C, H, W = logits.shape
        if (H * W) % 8:
            raise NotImplementedError('HIP path: H*W must be a multiple of 8')
        if target.dtype != torch.int64 or target.shape != (B, H, W):
            raise RuntimeError('target must be int64 of shape (B,H,W)')
        target = target.contiguous()
        dev = logits.device
        lse = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        acc = torch.zeros(2, dtype=torch.float64, device=dev)
        scal = torch.empty(2, dtype=torch.float32, device=dev)
        call('tss_cross_entropy_fwd', ptr(logits), ptr(target), ptr(lse), ptr(acc), ptr(scal[0:1]), ptr(scal[1:2]),
             B, C, H * W, int(ignore_index), N.dtype_code(logits.dtype), stream())
        ctx.ignore_index = int(ignore_index)
        ctx.save_for_backward(logits, target, lse, scal)
        return scal[0].clone()

    @staticmethod
    def backward(ctx, gout):
        logits, target, lse, scal = ctx.saved_tensors
        B, C, H, W = logits.shape
        gout = gout.to(torch.float32).contiguous()
        d = torch.empty_like(logits)
        call('tss_cross_entropy_bwd', ptr(logits), ptr(target), ptr(lse), ptr(scal[1:2]), ptr(gout), ptr(d),
             B, C, H * W, ctx.ignore_index, N.dtype_code(logits.dtype), stream())
        return d, None, None


def cross_entropy(logits, target, ignore_index=-100):
    """F.cross_entropy(logits, target, ignore_index=..., reduction='mean') for (B,C,H,W) logits."""
    return CrossEntropyFn.apply(logits, target, ignore_index)


class CrossEntropyLoss(torch.nn.Module):
    """Drop-in for nn.CrossEntropyLoss(ignore_index=...) as scripts/train_fastscnn.py:132 builds it."""

    def __init__(self, ignore_index=-100):
        super().__init__()
        self.ignore_index = ignore_index

    def forward(self, input, target):
        return cross_entropy(input, target, self.ignore_index)


class OHEMFn(Function):
    """ohem_loss (TSS/losses/ohem_loss.py:10-21): per-pixel CE, the (n+1)-th largest found by a device-side radix select."""
    _ws = {}

    @staticmethod
    def forward(ctx, logits, target, ignore_index, thresh_loss, numel_frac):
        _check_device(logits)
        logits = logits.contiguous()
        B, C, H, W = logits.shape
        if (H * W) % 8:
            raise NotImplementedError('HIP path: H*W must be a multiple of 8')
        if target.dtype != torch.int64 or target.shape != (B, H, W):
            raise RuntimeError('target must be int64 of shape (B,H,W)')
        target = target.contiguous()
        dev = logits.device
        key = (dev.type, dev.index)
        if key not in OHEMFn._ws:          # zeroed once; every call leaves it zeroed
            OHEMFn._ws[key] = torch.zeros(N.lib().tss_ohem_workspace_bytes(), dtype=torch.uint8, device=dev)
        lse = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        pix = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        out = torch.empty(5, dtype=torch.float32, device=dev)              # loss, then the 4 selection parameters
        n_top = int(B * H * W * float(numel_frac))
        call('tss_ohem_fwd', ptr(logits), ptr(target), ptr(lse), ptr(pix), ptr(OHEMFn._ws[key]), ptr(out[0:1]),
             ptr(out[1:5]), B, C, H * W, int(ignore_index), float(thresh_loss), n_top, N.dtype_code(logits.dtype),
             stream())
        ctx.ignore_index = int(ignore_index)
        ctx.save_for_backward(logits, target, lse, pix, out)
        return out[0].clone()

    @staticmethod
    def backward(ctx, gout):
        logits, target, lse, pix, out = ctx.saved_tensors
        B, C, H, W = logits.shape
        gout = gout.to(torch.float32).contiguous()
        d = torch.empty_like(logits)
        call('tss_ohem_bwd', ptr(logits), ptr(target), ptr(lse), ptr(pix), ptr(out[1:5]), ptr(gout), ptr(d),
             B, C, H * W, ctx.ignore_index, N.dtype_code(logits.dtype), stream())
        return d, None, None, None, None


def ohem_loss(input, target, ignore_index=-100, thresh_loss=0.35667494393873245, numel_frac=0.01):
    """TSS/losses/ohem_loss.py:10-21 (thresh_loss default = -log(0.7))."""
    return OHEMFn.apply(input, target, ignore_index, thresh_loss, numel_frac)


class OHEMLoss(torch.nn.Module):
    """Drop-in for TSS.losses.OHEMLoss (TSS/losses/ohem_loss.py:24-37), the loss of scripts/train_fastscnn.py."""

    def __init__(self, ignore_index=-100, thresh_loss=0.35667494393873245, numel_frac=0.01):
        super().__init__()
        self.ignore_index, self.thresh_loss, self.numel_frac = ignore_index, thresh_loss, numel_frac

    def forward(self, input, target):
        return ohem_loss(input, target, self.ignore_index, self.thresh_loss, self.numel_frac)


class UpsampleCrossEntropyFn(Function):
    """cross_entropy(F.interpolate(low, scale, bilinear, align_corners=True), target) without the full-res logits:
    one pass yields the loss and the unscaled low-res gradient (per-block tiles in a workspace, no atomics, nothing to
    zero-fill), backward gathers the tiles in a fixed order and scales by grad_out / #valid: bit-identical from run to run."""

    @staticmethod
    def forward(ctx, low, target, ho, wo, ignore_index):
        B, C, h, w = low.shape
        dev = low.device
        target = target.contiguous()
        nws = N.lib().tss_upsample_ce_ws(B, C, h, w, ho, wo)
        ws = torch.empty(nws, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)        # written by the kernel: no copy launch for the return value
        inv = torch.empty(1, dtype=torch.float32, device=dev)
        call('tss_upsample_ce_fwd', ptr(low), ld(low), ptr(target), ptr(ws), ptr(loss),
             ptr(inv), B, C, h, w, ho, wo, int(ignore_index), N.dtype_code(low.dtype), stream())
        ctx.geom = (B, C, h, w, ho, wo, ld(low), low.dtype)
        ctx.save_for_backward(ws, inv)
        return loss

    @staticmethod
    def backward(ctx, gout):
        ws, inv = ctx.saved_tensors
        B, C, h, w, ho, wo, ldl, dtype = ctx.geom
        gout = gout.to(torch.float32).contiguous()
        base = torch.empty((B, h, w, ldl), dtype=dtype, device=ws.device)   # pad channels are written as zeros
        call('tss_upsample_ce_bwd', ptr(ws), ptr(inv), ptr(gout), ptr(base), ldl, B, C, h, w, ho, wo,
             N.dtype_code(dtype), stream())
        return base.permute(0, 3, 1, 2)[:, :C], None, None, None, None


def upsample_cross_entropy(low, target, scale_factor=None, size=None, ignore_index=-100):
    """Fused decoder head + loss:
    F.cross_entropy(F.interpolate(low, scale_factor, mode='bilinear', align_corners=True), target, ignore_index)
    computed from the low-resolution logits; the full-resolution logits are never materialised."""
    low = to_nhwc(materialize(low))
    ho, wo = _out_size(low, size, scale_factor)
    if low.shape[1] > 24 or ho < low.shape[2] or wo < low.shape[3]:
        # outside the one-pass kernel's envelope (class registers, upsampling only): same result, unfused
        return cross_entropy(upsample_logits(low, size=(ho, wo)), target, ignore_index=ignore_index)
    if target.dtype != torch.int64 or tuple(target.shape) != (low.shape[0], ho, wo):
        raise RuntimeError('target must be int64 of shape (B,H,W) = %s' % ((low.shape[0], ho, wo),))
    _check_device(target)
    return UpsampleCrossEntropyFn.apply(low, target, ho, wo, ignore_index)


class UpsampleOHEMFn(Function):
    """ohem_loss(F.interpolate(low, scale, bilinear, align_corners=True), target) (TSS/losses/ohem_loss.py:10-21 on the
    decoder head of TSS/models/fastscnn.py:40-43) without the full-resolution logits: the per-pixel cross-entropy is computed
    straight from the low-res logits, the radix select runs on that [B,H,W] f32 array, and backward recomputes the softmax of
    the selected pixels from the low-res logits again (gradient tiles, fixed-order gather: no atomics)."""

    @staticmethod
    def forward(ctx, low, target, ho, wo, ignore_index, thresh_loss, numel_frac):
        B, C, h, w = low.shape
        dev = low.device
        target = target.contiguous()
        key = (dev.type, dev.index)
        if key not in OHEMFn._ws:          # zeroed once; every call leaves it zeroed
            OHEMFn._ws[key] = torch.zeros(N.lib().tss_ohem_workspace_bytes(), dtype=torch.uint8, device=dev)
        pix = torch.empty((B, ho, wo), dtype=torch.float32, device=dev)
        out = torch.empty(8, dtype=torch.float32, device=dev)              # loss, 3 pad, the 4 selection parameters (16 B aligned)
        code = N.dtype_code(low.dtype)
        call('tss_upsample_pixel_ce', ptr(low), ld(low), ptr(target), ptr(pix), B, C, h, w, ho, wo, int(ignore_index), code, stream())
        call('tss_ohem_select', ptr(pix), ptr(OHEMFn._ws[key]), ptr(out[0:1]), ptr(out[4:8]), B * ho * wo, float(thresh_loss),
             int(B * ho * wo * float(numel_frac)), stream())
        ctx.geom = (B, C, h, w, ho, wo, ld(low), low.dtype, int(ignore_index))
        ctx.save_for_backward(low, target, pix, out)
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        low, target, pix, out = ctx.saved_tensors
        B, C, h, w, ho, wo, ldl, dtype, ignore_index = ctx.geom
        dev = low.device
        gout = gout.to(torch.float32).contiguous()
        ws = torch.empty(N.lib().tss_upsample_ce_ws(B, C, h, w, ho, wo), dtype=torch.float32, device=dev)
        code = N.dtype_code(dtype)
        call('tss_upsample_ohem_grad', ptr(low), ldl, ptr(target), ptr(pix), ptr(out[4:8]), ptr(ws), B, C, h, w, ho, wo,
             ignore_index, code, stream())
        base = torch.empty((B, h, w, ldl), dtype=dtype, device=dev)
        call('tss_upsample_ce_bwd', ptr(ws), ptr(_unit(dev)), ptr(gout), ptr(base), ldl, B, C, h, w, ho, wo, code, stream())
        return base.permute(0, 3, 1, 2)[:, :C], None, None, None, None, None, None


_UNIT = {}


def _unit(dev):
    key = (dev.type, dev.index)
    if key not in _UNIT:
        _UNIT[key] = torch.ones(1, dtype=torch.float32, device=dev)
    return _UNIT[key]


def upsample_ohem_loss(low, target, scale_factor=None, size=None, ignore_index=-100, thresh_loss=0.35667494393873245,
                       numel_frac=0.01):
    """Fused decoder head + the reference recipe's loss (scripts/train_fastscnn.py with TSS/losses/ohem_loss.py):
    ohem_loss(F.interpolate(low, scale_factor, mode='bilinear', align_corners=True), target, ...) from the low-res logits."""
    low = to_nhwc(materialize(low))
    ho, wo = _out_size(low, size, scale_factor)
    B = low.shape[0]
    if low.shape[1] > 24 or ho < low.shape[2] or wo < low.shape[3] or (B * ho * wo) % 4:
        return ohem_loss(upsample_logits(low, size=(ho, wo)), target, ignore_index, thresh_loss, numel_frac)
    if target.dtype != torch.int64 or tuple(target.shape) != (B, ho, wo):
        raise RuntimeError('target must be int64 of shape (B,H,W) = %s' % ((B, ho, wo),))
    _check_device(target)
    return UpsampleOHEMFn.apply(low, target, ho, wo, ignore_index, thresh_loss, numel_frac)


def upsample_argmax_confusion(low, target=None, scale_factor=None, size=None, ignore_index=255, confusion=None,
                              want_pred=True):
    """argmax_confusion(F.interpolate(low, scale, bilinear, align_corners=True), ...) without the full-resolution logits
    (the evaluator's decoder head + metric update as one operator).  Returns (pred uint8 or None, confusion)."""
    low = to_nhwc(materialize(low))
    ho, wo = _out_size(low, size, scale_factor)
    B, C, h, w = low.shape
    if C > 24 or ho < h or wo < w:
        return argmax_confusion(upsample_logits(low, size=(ho, wo)), target, ignore_index=ignore_index,
                                confusion=confusion, want_pred=want_pred)
    pred = torch.empty((B, ho, wo), dtype=torch.uint8, device=low.device) if want_pred else None
    if target is not None:
        if target.dtype != torch.int64 or tuple(target.shape) != (B, ho, wo):
            raise RuntimeError('target must be int64 of shape (B,H,W) = %s' % ((B, ho, wo),))
        _check_device(target)
        if confusion is None:
            confusion = torch.zeros((C, C), dtype=torch.int64, device=low.device)
    call('tss_upsample_argmax_confusion', ptr(low), ld(low), ptr(target.contiguous()) if target is not None else None,
         ptr(pred), ptr(confusion), B, C, h, w, ho, wo, int(ignore_index), N.dtype_code(low.dtype), stream())
    return pred, confusion


def argmax_confusion(logits, target=None, num_classes=None, ignore_index=255, confusion=None, want_pred=True):
    """argmax over dim 1 (lowest index wins ties) and, with a target, the confusion-matrix update
    (rows = truth, cols = prediction) that create_segmentation_evaluator's metrics need (TSS/engine.py:65-72)."""
    _check_device(logits)
    logits = logits.contiguous()
    B, C, H, W = logits.shape
    pred = torch.empty((B, H, W), dtype=torch.uint8, device=logits.device) if want_pred else None
    if target is not None and confusion is None:
        confusion = torch.zeros((C, C), dtype=torch.int64, device=logits.device)
    call('tss_argmax_confusion', ptr(logits), ptr(target.contiguous()) if target is not None else None, ptr(pred),
         ptr(confusion), B, C, H * W, int(ignore_index), N.dtype_code(logits.dtype), stream())
    return pred, confusion
