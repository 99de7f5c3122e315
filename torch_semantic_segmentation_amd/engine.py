"""Train / eval step of the hot path: the counterpart of TSS/engine.py (rows R, S of SURVEY.md §8a).

`create_segmentation_trainer` reproduces the body of the reference's `update_fn` (TSS/engine.py:24-39):
train mode, zero_grad, forward, loss, backward, optimizer step, loss value.  ignite and apex are not
dependencies: the returned :class:`Trainer` is a plain object with `update(batch)` and `run(loader, epochs)`.

MI355X specifics
  * parameters and gradients of the model live in two flat f32 buffers (`FlatAdamW`), so the optimizer is one
    fused kernel (tss_adamw_step) and data-parallel training needs ONE RCCL all-reduce per step;
  * weight-gradient kernels accumulate straight into the flat gradient buffer (ops.direct_grads), so there is
    no per-parameter `.grad +=` launch;
  * zero_grad + forward + loss + backward can be captured once into a HIP graph (`use_graph=True`) and
    replayed, removing ~10^3 Python-side launches per step;
  * one process per GPU; gradients are averaged with `torch.distributed.all_reduce` (backend "nccl" = RCCL over
    xGMI on GPUs, "gloo" on CPU for tests).  BatchNorm statistics stay per replica unless the model
    went through `convert_syncbn_model` (ops.SyncBatchNorm: one small all-reduce per BatchNorm layer and direction).
"""
import os
from functools import partial

import torch
import torch.distributed as dist
from torch import nn

from . import _native as N
from . import ops


# ----------------------------------------------------------------------------- flat parameters + fused AdamW

class FlatAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction) on ONE flat f32 buffer.

    `param_groups[0]['lr']` stays a Python float so LR schedulers work; it is mirrored into a device scalar
    before every step (outside any captured graph).
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        params = [p for p in params]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise NotImplementedError('FlatAdamW: a single parameter group')
        ps = self.param_groups[0]['params']
        dev = ps[0].device
        if any(p.dtype != torch.float32 or p.device != dev for p in ps):
            raise TypeError('FlatAdamW: float32 parameters on one device')
        n = sum(p.numel() for p in ps)
        self.flat_param = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in ps:
                k = p.numel()
                self.flat_param[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat_param[off:off + k].view_as(p)
                p.grad = self.flat_grad[off:off + k].view_as(p)
                off += k
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        # step counter and learning rate: on the host (kernel arguments of ONE launch per step) unless device_state=True, which
        # keeps them in device memory behind a tick kernel so that step() itself can be captured in a HIP graph
        self.device_state = False
        self.step_count = 0
        self.state_vec = torch.zeros(3, dtype=torch.float32, device=dev)  # device_state: step, bias corrections
        self.lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=dev)
        self.grad_scale = 1.0

    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()

    def _check_aliases(self):
        """Every p.grad must still be its slice of flat_grad (model.zero_grad(set_to_none=True) or an optimizer-external
        `p.grad = ...` detaches it; stepping would then silently apply zero gradients)."""
        off = 0
        base = self.flat_grad.data_ptr()
        esz = self.flat_grad.element_size()
        for p in self.param_groups[0]['params']:
            g = p.grad
            if g is None or g.data_ptr() != base + esz * off:
                raise RuntimeError('FlatAdamW: a parameter gradient no longer aliases the flat gradient buffer (was '
                                   'model.zero_grad(set_to_none=True) called?); use optimizer.zero_grad() or reattach()')
            off += p.numel()

    def reattach(self):
        """Point every p.grad back at its slice of the flat gradient buffer."""
        off = 0
        for p in self.param_groups[0]['params']:
            k = p.numel()
            p.grad = self.flat_grad[off:off + k].view_as(p)
            off += k

    def state_dict(self):
        """torch.optim state_dict plus the flat moments and the device-side step counter (checkpoint / resume)."""
        sd = super().state_dict()
        sv = self.state_vec.clone()
        if not self.device_state:
            b1, b2 = self.param_groups[0]['betas']
            k = float(self.step_count)
            sv = torch.tensor([k, 1.0 - b1 ** k, (1.0 - b2 ** k) ** 0.5], dtype=torch.float32, device=sv.device)
        sd['flat'] = {'exp_avg': self.exp_avg.clone(), 'exp_avg_sq': self.exp_avg_sq.clone(), 'state_vec': sv}
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        flat = state_dict.pop('flat', None)
        if flat is None and state_dict.get('state'):
            # a torch.optim.AdamW checkpoint (per-parameter exp_avg / exp_avg_sq / step): its moments would land in self.state and
            # never be read -- a resume would silently restart the bias correction
            raise ValueError('FlatAdamW.load_state_dict: this state_dict has per-parameter state but no flat moments (it was not '
                             'written by FlatAdamW.state_dict()); pack it into the flat buffers before loading')
        super().load_state_dict(state_dict)
        if flat is not None:
            with torch.no_grad():
                self.exp_avg.copy_(flat['exp_avg'])
                self.exp_avg_sq.copy_(flat['exp_avg_sq'])
                self.state_vec.copy_(flat['state_vec'])
            self.step_count = int(round(float(flat['state_vec'][0])))

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        if not self.flat_param.is_cuda:
            raise RuntimeError('FlatAdamW runs on the HIP path only (no CPU fallback)')
        self._check_aliases()
        b1, b2 = g['betas']
        self.step_count += 1
        if self.device_state:
            self.lr_dev.fill_(float(g['lr']))
        N.call('tss_adamw_step', N.ptr(self.flat_param), N.ptr(self.flat_grad), N.ptr(self.exp_avg),
               N.ptr(self.exp_avg_sq), self.flat_param.numel(), N.ptr(self.lr_dev) if self.device_state else None,
               float(b1), float(b2), float(g['eps']), float(g['weight_decay']),
               N.ptr(self.state_vec) if self.device_state else None, float(self.grad_scale), float(g['lr']),
               int(self.step_count), N.stream())


# ----------------------------------------------------------------------------- distributed helpers

def setup_distributed(enable=True, local_rank=None, backend=None):
    """TSS/utils/training.py:5-19 without the GPU-only assumption: returns (world_size, rank, local_rank).
    Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run)."""
    if not enable:
        return 1, 0, 0
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0')) if local_rank is None else local_rank
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    launched = 'RANK' in os.environ and 'MASTER_ADDR' in os.environ     # under torch.distributed.run, even with 1 rank
    if (world > 1 or launched) and not dist.is_initialized():
        dist.init_process_group(backend or ('nccl' if torch.cuda.is_available() else 'gloo'), init_method='env://')
    if dist.is_initialized():
        return dist.get_world_size(), dist.get_rank(), local_rank
    return 1, 0, local_rank


def shard_batch(n_items, world_size, rank):
    """DistributedSampler's partition without shuffling (TSS/utils/training.py:54-61): rank r owns r, r+W, ..."""
    return list(range(rank, n_items, world_size))


def allreduce_mean_(flat, world_size, group=None):
    """ONE collective per step over the flat gradient buffer: sum here, the 1/world factor is folded into AdamW."""
    if world_size > 1 or (dist.is_available() and dist.is_initialized()):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


# ----------------------------------------------------------------------------- trainer

class Trainer:
    def __init__(self, model, optimizer, loss_fn, device=None, use_graph=False, world_size=None,
                 non_blocking=True, fuse_head_loss=True, graph_allreduce=None):
        self.model, self.optimizer, self.loss_fn = model, optimizer, loss_fn
        self.device = device
        self.non_blocking = non_blocking
        # gradients are SUMMED over the ranks and scaled by 1/world_size: the factor must be the size of the group the
        # all-reduce runs over, so it is derived from torch.distributed unless given (and checked when given)
        live = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if world_size is None:
            world_size = live
        elif world_size != live:
            raise ValueError('Trainer(world_size=%d) but torch.distributed has %d rank(s): gradients would be scaled wrongly'
                             % (world_size, live))
        self.world_size = world_size
        self.use_graph = use_graph
        self.flat = isinstance(optimizer, FlatAdamW)
        # model(x) followed by CrossEntropyLoss == the fused head+loss operator on model.forward_lowres(x): same
        # value and gradients, but the full-resolution logits never exist (-1.3 GB and ~0.8 ms at 8x1024x2048:
        # one 0.26 ms kernel instead of five that move 3.8 GB).  Only taken when nothing can observe the difference
        # (our loss class, a model that offers forward_lowres, no hooks on the model itself).
        self.fuse_head_loss = bool(fuse_head_loss) and type(loss_fn) in (ops.CrossEntropyLoss, ops.OHEMLoss) \
            and hasattr(model, 'forward_lowres') and hasattr(model, 'logit_scale')
        if self.flat:
            optimizer.grad_scale = 1.0 / world_size
        self.shadows = None
        # The gradient all-reduce INSIDE the captured step (RCCL collectives on the capturing stream are recorded into the HIP
        # graph): the step is then replay + optimizer, nothing else on the host.  Opt-in (argument or TSS_GRAPH_ALLREDUCE=1):
        # validated on hardware with a one-rank RCCL group only -- a multi-GPU node was never available to this repo -- so the
        # default keeps the collective an ordinary eager launch behind the replay, which is the form the 8-GPU run depends on.
        if graph_allreduce is None:
            graph_allreduce = os.environ.get('TSS_GRAPH_ALLREDUCE') == '1'
        self.graph_allreduce = bool(graph_allreduce) and self.flat and dist.is_available() and dist.is_initialized() \
            and dist.get_backend() == 'nccl'
        self._reduce_captured = False
        self._collectives_in_graph = False
        # captured steps, one per input slot: slot -> (graph, loss tensor).  Slot 0 is the ordinary one; a loader that
        # double-buffers its batches (HostBatchPipeline) captures one step per staging slot so that the step reads the
        # staged batch in place -- no device-to-device copy between the H2D copy and the step
        self._graphs = {}
        self._statics = {}
        self._one = None
        self._next_slot = 1           # slot 0 is the trainer's own; new_slots() hands out the rest, never twice
        self._pool = None             # one private memory pool shared by the captured steps of all slots (they never run concurrently)
        self.iteration = 0
        self.last_loss = None

    @property
    def _graph(self):
        g = self._graphs.get(0)
        return g[0] if g else None

    # -- one un-captured iteration: exactly the statements of the reference's update_fn
    def _forward_backward(self, x, y, prologue=None):
        if prologue is not None:      # device-side preparation of (x, y), e.g. the uint8 decode: part of the captured step
            prologue()
        self.model.train()
        if self.shadows is None:
            self.shadows = ops.WeightShadows(self.model)
        # the optimizer step of the previous iteration changed the weights: one launch rewrites their bf16 shadows AND clears the
        # flat gradient buffer (optimizer.zero_grad() of TSS/engine.py:28)
        if self.flat and os.environ.get('TSS_MEMSET_NODES') == '1':
            # diagnostic only (tools/graph_memset_probe.py): the gradient buffer cleared by a memset NODE
            g = self.optimizer.flat_grad
            N.call('tss_memset_zero', N.ptr(g), g.numel() * g.element_size(), N.stream())
            self.shadows.refresh()
        elif not self.shadows.refresh(zero=self.optimizer.flat_grad if self.flat else None):
            self.optimizer.zero_grad()
        if self._one is None or self._one.device != x.device:
            self._one = torch.ones((), dtype=torch.float32, device=x.device)     # d(loss)/d(loss): no fill launch per step
        with ops.direct_grads(self.flat), self.shadows:
            if self.fuse_head_loss and not (self.model._forward_hooks or self.model._forward_pre_hooks):
                low = self.model.forward_lowres(x)
                if type(self.loss_fn) is ops.OHEMLoss:
                    loss = ops.upsample_ohem_loss(low, y, scale_factor=self.model.logit_scale, ignore_index=self.loss_fn.ignore_index,
                                                  thresh_loss=self.loss_fn.thresh_loss, numel_frac=self.loss_fn.numel_frac)
                else:
                    loss = ops.upsample_cross_entropy(low, y, scale_factor=self.model.logit_scale,
                                                      ignore_index=self.loss_fn.ignore_index)
            else:
                y_pred = self.model(x)
                loss = self.loss_fn(y_pred, y)
            loss.backward(self._one if (loss.dtype == torch.float32 and loss.dim() == 0 and loss.device == self._one.device) else None)
        return loss

    def _reduce_and_step(self, captured_reduce=False):
        if captured_reduce:
            pass                # the all-reduce of the flat gradient buffer was replayed with the step
        elif self.world_size > 1 or (dist.is_available() and dist.is_initialized()):
            if self.flat:
                allreduce_mean_(self.optimizer.flat_grad, self.world_size)
            else:
                for p in self.model.parameters():
                    if p.grad is not None:
                        dist.all_reduce(p.grad)
                        p.grad.div_(self.world_size)
        self.optimizer.step()

    def new_slots(self, n):
        """`n` fresh input-slot numbers (for a loader that captures one step per staging buffer).  Slot numbers are never
        reused, so a pipeline created after another one was dropped can not inherit its captured steps."""
        first = self._next_slot
        self._next_slot += int(n)
        return list(range(first, first + int(n)))

    def release_slot(self, slot):
        """Drop the captured step and the static input buffers of `slot` (its activations go back to the shared pool)."""
        self._graphs.pop(slot, None)
        self._statics.pop(slot, None)

    def static_batch(self, x, y, slot=0, adopt=False):
        """The (image, target) device buffers the captured step of `slot` reads.  A loader can fill them in place (H2D copy
        straight into them) and pass them to step_async, which then skips its own device-to-device copy.  adopt=True:
        x and y (device tensors) BECOME the static buffers of a slot that has none yet; for a slot that already has
        buffers they must BE those buffers (a captured step reads fixed addresses)."""
        if adopt and slot in self._statics and x.is_cuda and y.is_cuda:
            sx, sy = self._statics[slot]
            if sx.data_ptr() != x.data_ptr() or sy.data_ptr() != y.data_ptr():
                raise RuntimeError('Trainer.static_batch(adopt=True): slot %d already has other input buffers; take fresh '
                                   'slot numbers from Trainer.new_slots() or release_slot() it first' % slot)
        if slot not in self._statics:
            dev = self.device if self.device is not None else x.device
            if adopt and x.is_cuda and y.is_cuda:
                self._statics[slot] = (x, y)
            else:
                self._statics[slot] = (torch.empty(x.shape, dtype=x.dtype, device=dev), torch.empty(y.shape, dtype=y.dtype, device=dev))
        sx, sy = self._statics[slot]
        if tuple(sx.shape) != tuple(x.shape) or tuple(sy.shape) != tuple(y.shape) or sx.dtype != x.dtype or sy.dtype != y.dtype:
            raise ValueError('HIP-graph trainer: the batch shape/dtype is fixed at the first step '
                             f'({tuple(sx.shape)} {sx.dtype}); got {tuple(x.shape)} {x.dtype}')
        if x.data_ptr() != sx.data_ptr():
            sx.copy_(x, non_blocking=True)
        if y.data_ptr() != sy.data_ptr():
            sy.copy_(y, non_blocking=True)
        return sx, sy

    def _capture(self, x, y, slot=0, prologue=None):
        sx, sy = self.static_batch(x, y, slot)
        # warm-up on a side stream (lazily initialised state must exist before capture); buffers that a
        # forward pass mutates (BatchNorm running statistics) are restored so the warm-up leaves no trace
        saved = [(b, b.clone()) for b in self.model.buffers()]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self._forward_backward(sx, sy, prologue)
        torch.cuda.current_stream().wait_stream(side)
        with torch.no_grad():
            for b, v in saved:
                b.copy_(v)
        dot = getattr(self, 'debug_dot', None)       # tools/graph_memset_probe.py: hipGraphDebugDotPrint of the captured step
        graph = torch.cuda.CUDAGraph(keep_graph=True) if dot else torch.cuda.CUDAGraph()
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        # with a live RCCL process group its watchdog thread polls events while we capture: in the default 'global' capture
        # mode that aborts the capture (seen as a flaky crash); only this thread's calls are checked in 'thread_local' mode
        mode = 'thread_local' if (dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl') else 'global'
        with torch.cuda.graph(graph, pool=self._pool, capture_error_mode=mode):
            loss = self._forward_backward(sx, sy, prologue).detach()
            if self.graph_allreduce:
                allreduce_mean_(self.optimizer.flat_grad, self.world_size)
        self._graphs[slot] = (graph, loss)
        self._reduce_captured = self.graph_allreduce
        if dot:
            try:        # the captured hipGraph_t straight to the runtime's own printer
                import ctypes
                hip = ctypes.CDLL('libamdhip64.so')
                hip.hipGraphDebugDotPrint.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint]
                rc = hip.hipGraphDebugDotPrint(ctypes.c_void_p(graph.raw_cuda_graph()), dot.encode(), 1)   # 1 = verbose
                if rc != 0:
                    raise RuntimeError('hipGraphDebugDotPrint returned %d' % rc)
            except Exception as exc:      # noqa: BLE001
                import warnings
                warnings.warn('graph dot dump failed: %s' % exc)

    def step_async(self, x, y, slot=0, prologue=None):
        """One training iteration; returns the loss as a device tensor (no host sync).  `slot` selects one of several
        captured steps (each with its own input buffers, see static_batch); `prologue` is a callable that prepares (x, y)
        on the device and is captured with the step."""
        if self.use_graph and not self._graphs and any(ops._sync_group(m, any_mode=True) is not None for m in self.model.modules()
                                                          if isinstance(m, nn.modules.batchnorm._BatchNorm)):
            # cross-replica BatchNorm puts 2 small collectives per BatchNorm layer inside the step.  RCCL collectives on the
            # capturing stream are recorded into the HIP graph like kernels (validated with a one-rank RCCL group on one
            # MI355X, tests/test_gpu_models.py; a multi-GPU node was not available); collectives of a host-side backend
            # (gloo) cannot be captured, and if the capture fails for any reason the step runs un-captured (correct,
            # ~10^3 host launches per step).
            import warnings
            backend = dist.get_backend() if dist.is_initialized() else None
            dev = next(self.model.parameters()).device
            groups = {id(g): g for g in (ops._sync_group(m, any_mode=True) for m in self.model.modules()
                                         if isinstance(m, nn.modules.batchnorm._BatchNorm)) if g is not None}
            if groups and all(ops._exchange(g, dev) is not None for g in groups.values()):
                # round 4: the statistics cross the ranks INSIDE the finalize kernels (mailboxes over HIP IPC, csrc/xchg.hip): the step
                # contains no collective at all and is captured like any other, whatever the backend of the process group
                pass
            elif backend != 'nccl':
                warnings.warn('SyncBatchNorm over the %s backend: the training step is not captured in a HIP graph' % backend)
                self.use_graph = False
            else:
                self._collectives_in_graph = True
        if self.use_graph and slot not in self._graphs:
            if self._collectives_in_graph or self.graph_allreduce:
                # a step with RCCL collectives inside: if ANY slot's capture fails, every slot runs un-captured from then on
                # (correct, ~10^3 host launches per step) -- one decision per trainer, taken by whichever slot captures first
                try:
                    self._capture(x, y, slot, prologue)
                except Exception as exc:      # noqa: BLE001 -- any capture failure: fall back, loudly
                    import warnings
                    warnings.warn('HIP-graph capture of a step with RCCL collectives failed (%s: %s); running un-captured'
                                  % (type(exc).__name__, exc))
                    self._graphs.clear()
                    self.use_graph = False
                    self._reduce_captured = False
                    torch.cuda.synchronize()
            else:
                self._capture(x, y, slot, prologue)
        elif self.use_graph:
            self.static_batch(x, y, slot)
        if self.use_graph:
            graph, loss = self._graphs[slot]
            graph.replay()
        else:
            loss = self._forward_backward(x, y, prologue).detach()
        self._reduce_and_step(captured_reduce=self.use_graph and self._reduce_captured)
        self.iteration += 1
        return loss

    def update(self, batch):
        """TSS/engine.py:24-39: returns loss.item() (a device->host sync per iteration, as in the reference)."""
        x, y = batch
        if self.device is not None and not self.use_graph:   # graph mode copies straight into the static buffers
            x = x.to(self.device, non_blocking=self.non_blocking)
            y = y.to(self.device, non_blocking=self.non_blocking)
        self.last_loss = self.step_async(x, y).item()
        return self.last_loss

    def run(self, data, max_epochs=1):
        history = []
        for _ in range(max_epochs):
            for batch in data:
                history.append(self.update(batch))
        return history


class HostBatchPipeline:
    """The host side of `update_fn` under load (TSS/engine.py:27: `x.to(device, non_blocking=True)` every iteration): the
    batch of step i+1 crosses PCIe on a copy stream while step i computes, through `depth` device staging slots.  With a
    captured trainer there is one captured step per slot, reading its slot in place: nothing but the graph replay (and the
    optimizer) is launched per step.

    wire = 'f32': float32 NCHW image + int64 target, as the reference's DataLoader delivers them (335 MB per 8 x 3 x 1024 x
           2048 batch -- 6.1 ms over PCIe Gen5, as long as the step itself; hidden, but only just);
    wire = 'u8' : uint8 image (CHW, or HWC as decoded: image_hwc=True) + uint8 target (67 MB); `mean` / `std` are the
           albumentations.Normalize constants (scripts/train_fastscnn.py:62-68), applied on the device by tss_decode_batch_u8
           at the top of the (captured) step.

        pipe = HostBatchPipeline(trainer, example_x_f32, example_y_i64, wire='u8', mean=..., std=...)
        pipe.put(x0, y0)
        for next_batch in loader:          # pinned host tensors (DataLoader(pin_memory=True)); others are pinned by a copy
            pipe.put(*next_batch)          # H2D of the next batch starts now, on the copy stream
            loss = pipe.step()             # waits for the oldest staged batch only, runs the step on it
    """

    def __init__(self, trainer, example_x, example_y, wire='f32', mean=None, std=None, image_hwc=False, depth=2, device=None):
        if wire not in ('f32', 'u8'):
            raise ValueError("wire must be 'f32' or 'u8'")
        self.trainer, self.wire, self.image_hwc, self.depth = trainer, wire, bool(image_hwc), int(depth)
        dev = device if device is not None else (trainer.device if trainer.device is not None else torch.device('cuda', torch.cuda.current_device()))
        self.device = torch.device(dev)
        B, C, H, W = example_x.shape
        self.geom = (B, C, H, W)
        f32 = lambda: (torch.empty((B, C, H, W), dtype=torch.float32, device=self.device),      # noqa: E731
                       torch.empty((B, H, W), dtype=torch.int64, device=self.device))
        if wire == 'u8':
            import ctypes
            xs = (B, H, W, C) if image_hwc else (B, C, H, W)
            self.stage = [(torch.empty(xs, dtype=torch.uint8, device=self.device), torch.empty((B, H, W), dtype=torch.uint8, device=self.device))
                          for _ in range(depth)]
            self._mean = (ctypes.c_float * 3)(*([float(v) for v in mean] + [0.0] * 3)[:3]) if mean is not None else None
            self._std = (ctypes.c_float * 3)(*([float(v) for v in std] + [1.0] * 3)[:3]) if std is not None else None
            self.decoded = f32()             # one decoded batch, shared by the slots: written and read inside one step
        else:
            self.stage = [f32() for _ in range(depth)]
            self.decoded = None
        self.slots = trainer.new_slots(depth)         # private slot numbers of this pipeline inside the trainer, never reused
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.ready = [torch.cuda.Event() for _ in range(depth)]
        self.consumed = [torch.cuda.Event() for _ in range(depth)]
        self._keep = [None] * depth          # host tensors stay referenced until their copy has been consumed
        self._head = self._count = 0

    def put(self, x, y):
        """Start the H2D copy of one batch.  Raises when all `depth` slots are still in use."""
        if self._count == self.depth:
            raise RuntimeError('HostBatchPipeline: %d batches already staged; call step() first' % self.depth)
        slot = (self._head + self._count) % self.depth
        sx, sy = self.stage[slot]
        if tuple(x.shape) != tuple(sx.shape) or x.dtype != sx.dtype or tuple(y.shape) != tuple(sy.shape) or y.dtype != sy.dtype:
            raise ValueError('HostBatchPipeline(wire=%r): expected image %s %s and target %s %s, got %s %s / %s %s'
                             % (self.wire, tuple(sx.shape), sx.dtype, tuple(sy.shape), sy.dtype, tuple(x.shape), x.dtype,
                                tuple(y.shape), y.dtype))
        if not x.is_cuda and not x.is_pinned():
            x = x.pin_memory()
        if not y.is_cuda and not y.is_pinned():
            y = y.pin_memory()
        self._keep[slot] = (x, y)
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(self.consumed[slot])      # the step that read this slot last has finished with it
            sx.copy_(x, non_blocking=True)
            sy.copy_(y, non_blocking=True)
            self.ready[slot].record(self.copy_stream)
        self._count += 1

    def close(self):
        """Release this pipeline's captured steps and input slots in the trainer (call when the loader is re-created, e.g. per
        epoch); the pipeline can not be used afterwards."""
        torch.cuda.current_stream(self.device).synchronize()
        self.copy_stream.synchronize()
        for s in self.slots:
            self.trainer.release_slot(s)
        self.slots, self.stage, self.decoded, self._keep, self._count = [], [], None, [], 0

    def _decode(self, slot):
        sx, sy = self.stage[slot]
        dx, dy = self.decoded
        B, C, H, W = self.geom
        N.call('tss_decode_batch_u8', N.ptr(sx), int(self.image_hwc), self._mean, self._std, N.ptr(dx), N.ptr(sy), N.ptr(dy),
               B, C, H * W, N.stream())

    def step(self):
        """Run one training step on the oldest staged batch; returns the loss as a device tensor."""
        if self._count == 0:
            raise RuntimeError('HostBatchPipeline.step(): nothing staged; call put() first')
        slot = self._head
        main = torch.cuda.current_stream(self.device)
        main.wait_event(self.ready[slot])
        tslot = self.slots[slot]
        if self.wire == 'u8':
            dx, dy = self.decoded
            self.trainer.static_batch(dx, dy, tslot, adopt=True)
            loss = self.trainer.step_async(dx, dy, slot=tslot, prologue=lambda: self._decode(slot))
        else:
            sx, sy = self.stage[slot]
            self.trainer.static_batch(sx, sy, tslot, adopt=True)
            loss = self.trainer.step_async(sx, sy, slot=tslot)
        self.consumed[slot].record(main)
        self._head = (self._head + 1) % self.depth
        self._count -= 1
        return loss


def create_segmentation_trainer(model, optimizer, loss_fn, device, use_f16=False, logging=True,
                                non_blocking=True, use_graph=False, world_size=None, fuse_head_loss=True):
    """Same arguments as TSS/engine.py:22.  `use_f16` selects bf16 activations (f32 master parameters), the
    MI355X counterpart of the reference's apex amp O2 branch (TSS/engine.py:32-34); no loss scaling is needed."""
    from .models import set_compute_dtype
    set_compute_dtype(model, torch.bfloat16 if use_f16 else torch.float32)
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    return Trainer(model, optimizer, loss_fn, device=device, use_graph=use_graph, world_size=world_size,
                   non_blocking=non_blocking, fuse_head_loss=fuse_head_loss)


# ----------------------------------------------------------------------------- evaluator

class EvalPrep:
    """Model-wide preparation of an eval-mode forward: bf16 shadows of the 1x1 weights (ops.WeightShadows, one launch), the affines
    of every BatchNorm that runs on its running statistics (ops.EvalAffines, one launch) and the bf16 tap-major copies of the dense
    3x3 weights (ops.Dense3x3Shadows, one small launch per layer).  `with prep:` refreshes them from the live parameters / buffers
    and makes them visible to the layers inside; with `frozen=True` they are refreshed by refresh() only (weights that do not change
    between forwards: the preparation leaves the forward, as `model.half()` leaves the timed loop of TSS/utils/benchmark.py)."""

    def __init__(self, model, frozen=False):
        self.shadows, self.affines, self.w3 = ops.WeightShadows(model), ops.EvalAffines(model), ops.Dense3x3Shadows(model)
        self.frozen, self.fresh = frozen, False

    def refresh(self):
        self.shadows.refresh()
        self.affines.refresh()
        self.w3.refresh()
        self.fresh = True

    def __enter__(self):
        if not (self.frozen and self.fresh):
            self.refresh()
        self.shadows.__enter__()
        self.affines.__enter__()
        self.w3.__enter__()
        return self

    def __exit__(self, *exc):
        self.w3.__exit__(*exc)
        self.affines.__exit__(*exc)
        self.shadows.__exit__(*exc)
        return False


class Evaluator:
    """model.eval() + no_grad forward, argmax, confusion matrix -> IoU / mIoU / accuracy / Dice
    (the metrics of create_segmentation_evaluator, TSS/engine.py:59-82; confusion rows = truth)."""

    def __init__(self, model, device=None, num_classes=19, loss_fn=None, ignore_index=255, non_blocking=True):
        self.model, self.device, self.num_classes = model, device, num_classes
        self.loss_fn, self.ignore_index, self.non_blocking = loss_fn, ignore_index, non_blocking

    @torch.no_grad()
    def run(self, data):
        self.model.eval()
        cm = None
        loss_sum, n = 0.0, 0
        prep = None
        for x, y in data:
            if self.device is not None:
                x = x.to(self.device, non_blocking=self.non_blocking)
                y = y.to(self.device, non_blocking=self.non_blocking)
            if prep is None:
                prep = EvalPrep(self.model)
            with prep:
                cm, loss_sum, n = self._batch(x, y, cm, loss_sum, n)
        metrics = confusion_metrics(cm.cpu().double())
        if self.loss_fn is not None and n:
            metrics['loss'] = loss_sum / n
        return metrics

    def _batch(self, x, y, cm, loss_sum, n):
        fused = (self.loss_fn is None and hasattr(self.model, 'forward_lowres') and hasattr(self.model, 'logit_scale')
                 and not (self.model._forward_hooks or self.model._forward_pre_hooks))
        if fused:   # decoder upsample + argmax + confusion matrix as one operator: no full-resolution logits
            low = self.model.forward_lowres(x)
            _, cm = ops.upsample_argmax_confusion(low, y, scale_factor=self.model.logit_scale,
                                                  ignore_index=self.ignore_index, confusion=cm, want_pred=False)
            return cm, loss_sum, n
        logits = self.model(x)
        _, cm = ops.argmax_confusion(logits, y, ignore_index=self.ignore_index, confusion=cm, want_pred=False)
        if self.loss_fn is not None:
            loss_sum += float(self.loss_fn(logits, y)) * x.shape[0]
            n += x.shape[0]
        return cm, loss_sum, n


def confusion_metrics(cm):
    tp = cm.diag()
    iou = tp / (cm.sum(0) + cm.sum(1) - tp + 1e-15)
    return {'iou': iou, 'miou': iou.mean().item(), 'accuracy': (tp.sum() / (cm.sum() + 1e-15)).item(),
            'dice': 2 * tp / (cm.sum(0) + cm.sum(1) + 1e-15)}


def create_segmentation_evaluator(model, device, num_classes=19, loss_fn=None, non_blocking=True):
    return Evaluator(model, device, num_classes=num_classes, loss_fn=loss_fn, non_blocking=non_blocking)


# ----------------------------------------------------------------------------- deep supervision (row S)

class DeepSupervisionWrapper(nn.Module):
    """Forward hooks on inner blocks feed auxiliary heads in training mode
    (TSS/wrappers/deep_supervision_wrapper.py:10-43): train -> (output, [aux...]); eval -> output."""

    def __init__(self, module, auxiliary_modules):
        super().__init__()
        self.module = module
        self.layers = [layer for layer, _ in auxiliary_modules]
        self.auxiliary = nn.ModuleList([head for _, head in auxiliary_modules])

    def forward(self, input):
        if not self.training:
            return self.module(input)
        aux_outputs = [None] * len(self.layers)

        def capture(_module, _inputs, output, slot, head):
            aux_outputs[slot] = head(output)

        handles = [layer.register_forward_hook(partial(capture, slot=i, head=head))
                   for i, (layer, head) in enumerate(zip(self.layers, self.auxiliary))]
        try:
            output = self.module(input)
        finally:
            for h in handles:
                h.remove()
        return output, aux_outputs


# ----------------------------------------------------------------------------- inference (TSS/utils/benchmark.py)

class GraphedInference:
    """eval-mode, no-grad forward of `model` captured once in a HIP graph and replayed: the input is copied into a fixed
    device buffer, the returned logits live in a fixed output buffer (overwritten by the next call).  Shapes are fixed
    at the first call.  `lowres=True` returns the 1/8-resolution logits of `model.forward_lowres` (what the fused
    evaluation head consumes) instead of the x8-upsampled ones.
    `frozen_weights=False` (default): the model-wide weight / statistics preparation (EvalPrep) is captured with the layers, so every
    replay follows the live parameters and buffers.  `frozen_weights=True`: it runs once, before the capture, and again only when
    refresh_weights() is called -- for deployed models whose weights do not change between forwards."""

    def __init__(self, model, lowres=False, frozen_weights=False):
        self.model, self.lowres, self.frozen = model, lowres, bool(frozen_weights)
        self._graph = self._x = self._out = self._prep = None

    def refresh_weights(self):
        """Re-run the weight / statistics preparation after the parameters or buffers changed (frozen_weights=True only)."""
        if self._prep is not None:
            self._prep.refresh()

    def _forward(self, x):
        if self._prep is None:
            self._prep = EvalPrep(self.model, frozen=self.frozen)
        with self._prep:      # model-wide preparation launches: captured with the rest (live weights) unless frozen_weights
            return self.model.forward_lowres(x) if self.lowres else self.model(x)

    def static_input(self, x):
        """The device buffer the captured forward reads: fill it in place (H2D copy straight into it) and pass it to
        __call__, which then skips its own device-to-device copy."""
        if self._x is None:
            self._x = torch.empty_like(x)
        if x.data_ptr() != self._x.data_ptr():
            self._x.copy_(x, non_blocking=True)
        return self._x

    @torch.no_grad()
    def __call__(self, x):
        if self._graph is None:
            self.model.eval()
            self.static_input(x)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._forward(self._x)
            torch.cuda.current_stream().wait_stream(side)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._out = self._forward(self._x)
        elif tuple(x.shape) != tuple(self._x.shape) or x.dtype != self._x.dtype:
            raise ValueError(f'GraphedInference: input fixed at {tuple(self._x.shape)} {self._x.dtype}, got {tuple(x.shape)} {x.dtype}')
        if x.data_ptr() != self._x.data_ptr():
            self._x.copy_(x, non_blocking=True)
        self._graph.replay()
        return self._out


def benchmark_model(model, batch, iterations, warmup, use_graph=False):
    """Same arguments and result keys as TSS/utils/benchmark.py:6-33 (fps / min / max / mean / std of the forward time
    in seconds, gradients disabled).  Each iteration is bracketed by a device synchronisation -- kernels are
    asynchronous, the reference's host-side `time()` pair only measures a GPU model if something blocks.  The caller
    chooses train/eval mode as in the reference; `use_graph` replays a captured eval forward (GraphedInference)."""
    import time
    import numpy as np
    fwd = GraphedInference(model) if use_graph else model
    cuda = batch.is_cuda
    record = np.zeros(iterations)
    with torch.set_grad_enabled(False):
        for _ in range(warmup):
            fwd(batch)
        for it in range(iterations):
            if cuda:
                torch.cuda.synchronize()
            start = time.perf_counter()
            fwd(batch)
            if cuda:
                torch.cuda.synchronize()
            record[it] = time.perf_counter() - start
    return {'fps': 1 / np.mean(record), 'min': np.min(record), 'max': np.max(record), 'mean': np.mean(record),
            'std': np.std(record)}
