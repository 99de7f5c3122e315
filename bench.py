#!/usr/bin/env python
"""Headline benchmark: images/sec of one full FastSCNN train step (zero_grad + forward + CrossEntropy(ignore 255)
+ backward + AdamW) on synthetic 8 x 3 x 1024 x 2048 batches per GPU, bf16 activations / f32 parameters.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model fastscnn|contextnet14] [--dtype bf16|f32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     : the kernel with the largest share of step time, timed with HIP events on its launch stream
                 (libtss_hip's profiler) over a few un-captured steps; achieved = algorithmic bytes / time.
  cpu_baseline : the CPU oracle (oracle/, a torch restatement of the reference) timed on this node's host
                 cores on a bounded sample of the same workload, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# sum over the convolutions of (input + output elements) at 8 x 3 x 1024 x 2048 (fastscnn_aspp: the FastSCNN trunk
# without its classifier (1819 M - 307 M) + the ASPP head at 1/8 resolution: 4 x (128 + 128) + (640 + 128) + (128 + 128) + (128 + 19)
# channels x 262 k pixels = 576 M)
S_REF = {'fastscnn': 1819e6, 'contextnet14': 2086e6, 'fastscnn_aspp': 2088e6}
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
HBM_COPY_GBS = 6290.0        # ... and the measured float4 copy rate on MI355X (79 % of spec)
MFMA_BF16_PEAK_TF = 2500.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--model', default='fastscnn', choices=['fastscnn', 'contextnet12', 'contextnet14', 'contextnet18', 'fastscnn_aspp', 'lednet', 'esnet'])
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--mode', default='train', choices=['train', 'eval'],
                    help='eval = SURVEY config C5: eval-mode no-grad forward, default 1 x 3 x 2048 x 4096 (never the headline line)')
    ap.add_argument('--batch', type=int, default=None, help='images per GPU (default 8; 1 in eval mode)')
    ap.add_argument('--height', type=int, default=None)
    ap.add_argument('--width', type=int, default=None)
    ap.add_argument('--graph', default='auto', choices=['auto', 'on', 'off'])
    ap.add_argument('--fuse-head', default='on', choices=['on', 'off'],
                    help='decoder upsample + cross-entropy as one operator (same value and gradients)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    ap.add_argument('--host-batch', action='store_true',
                    help='also report the PCIe-inclusive rate: the batch is copied from pinned host memory every step '
                         '(never the headline value; printed to stderr)')
    ap.add_argument('--stock', action='store_true', help='also time the stock PyTorch-ROCm (MIOpen) path of the oracle modules')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the extra.contextnet14 / extra.eval_c5 legs (BASELINE configs 3 and 5; N=1, default workload only)')
    ap.add_argument('--syncbn', action='store_true',
                    help='convert_syncbn_model before training (cross-replica BatchNorm, SURVEY.md section 8f N1)')
    args = ap.parse_args()
    dflt = (8, 1024, 2048) if args.mode == 'train' else (1, 2048, 4096)
    args.batch, args.height, args.width = (args.batch or dflt[0], args.height or dflt[1], args.width or dflt[2])
    return args


def build_model(name):
    import torch_semantic_segmentation_amd as tssa
    from torch_semantic_segmentation_amd.models import fastscnn as _f  # noqa: F401
    import importlib
    F = importlib.import_module('torch_semantic_segmentation_amd.models.fastscnn')
    C = importlib.import_module('torch_semantic_segmentation_amd.models.contextnet')
    A = importlib.import_module('torch_semantic_segmentation_amd.models.aspp')
    L = importlib.import_module('torch_semantic_segmentation_amd.models.lednet')
    E = importlib.import_module('torch_semantic_segmentation_amd.models.esnet')
    ctor = {'esnet': E.ESNet, 'lednet': L.lednet, 'fastscnn': F.fastscnn, 'contextnet12': C.contextnet12, 'contextnet14': C.contextnet14,
            'contextnet18': C.contextnet18, 'fastscnn_aspp': A.fastscnn_aspp}[name]
    torch.manual_seed(0)
    return ctor(3, 19), tssa


def synthetic(batch, h, w, seed, device):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, 3, h, w, generator=g, dtype=torch.float32)
    y = torch.randint(0, 19, (batch, h, w), generator=g, dtype=torch.int64)
    y[torch.rand(batch, h, w, generator=g) < 0.05] = 255
    return x.to(device), y.to(device)


def algorithmic_step_bytes(model_name, batch, h, w, esz):
    """A_step = 3*S*b + 4*U*b + 8*T, S scaled from the 8x1024x2048 figures."""
    S_ref = S_REF.get(model_name)
    if S_ref is None:
        return None
    scale = batch * h * w / (8.0 * 1024 * 2048)
    U = batch * 19.0 * h * w
    T = batch * 1.0 * h * w
    return 3 * S_ref * scale * esz + 4 * U * esz + 8 * T


def pmc_traffic(symbol, args):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/*_pmc_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of this command, FETCH_SIZE doubled as the gfx950 note in
    MI355X_MICROARCH.md prescribes).  Counters cannot be read from inside the timed process, so the number is the
    recorded one for the default workload; any other workload reports null."""
    import glob
    import json
    if (args.model, args.batch, args.height, args.width, args.dtype) != ('fastscnn', 8, 1024, 2048, 'bf16'):
        return None, None
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', '*_pmc_traffic.json')))
    if not files:
        return None, None
    table = json.load(open(files[-1]))
    source = 'profiles/' + os.path.basename(files[-1]) + ' (committed rocprofv3 --pmc passes of this command; not read live)'
    tot = n = 0.0       # launch-weighted mean over every kernel the entry point dispatched to
    for alt in symbol.split('|'):
        for k, v in table.items():
            if k.startswith(alt.split('<')[0]) and (('<' not in alt) or k.startswith(alt.rstrip('>'))):
                tot += v['hbm_bytes_per_launch'] * v['launches_sampled']
                n += v['launches_sampled']
    return (round(tot / n), source) if n else (None, None)


def host_cores():
    """Threads the CPU baseline uses: the cores this process may run on (cgroup/affinity aware), at most 16 --
    a gpurun box exposes 256 logical CPUs but grants a 16-core share per GPU, and torch oversubscribed on 256
    threads is ~100x slower than on 16."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline(model_name, seconds):
    """Oracle (port) on the host cores: FastSCNN fwd+CE+bwd+AdamW at BASELINE config 1 (4x3x512x1024, f32)."""
    from oracle import nets
    from oracle.recipe import synthetic_batch, train_step
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = nets.build(model_name)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-5)
    loss_fn = torch.nn.CrossEntropyLoss(ignore_index=255)
    x, y = synthetic_batch(4, 512, 1024)
    train_step(m, opt, loss_fn, x, y)  # warm-up
    t0 = time.time()
    n = 0
    while n < 2 or (time.time() - t0 < seconds and n < 50):
        train_step(m, opt, loss_fn, x, y)
        n += 1
    dt = time.time() - t0
    return {'value': round(4 * n / dt, 3), 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'sample': '%d train steps of %s at 4x3x512x1024 f32 (BASELINE config 1), oracle/nets.py on torch CPU, %.1f s'
                      % (n, model_name, dt)}


def stock_gpu(model_name, batch, h, w, device, steps=5):
    """What you get without this project: the same modules on stock PyTorch-ROCm (MIOpen), channels_last + bf16 autocast."""
    from oracle import nets
    torch.manual_seed(0)
    m = nets.build(model_name).to(device).to(memory_format=torch.channels_last)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-5, fused=True)
    loss_fn = torch.nn.CrossEntropyLoss(ignore_index=255)
    x, y = synthetic(batch, h, w, 1234, device)
    x = x.contiguous(memory_format=torch.channels_last)

    def step():
        m.train()
        opt.zero_grad()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out = m(x)
        loss = loss_fn(out.float(), y)
        loss.backward()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return batch * steps / (time.time() - t0)


def timed_steps(step_fn, steps, device, barrier=None, chunk=10):
    """Time EXACTLY `steps` calls of step_fn between two (barrier + synchronize) pairs; besides the wall time of the
    whole region, HIP events on the launch stream every `chunk` steps give per-chunk times without any host sync inside
    the region (median / min of those: SURVEY.md section 8d asks for the median of >= 50 steps)."""
    torch.cuda.synchronize()
    if barrier:
        barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True)]
    t0 = time.perf_counter()
    marks[0].record()
    out = None
    for i in range(steps):
        out = step_fn()
        if (i + 1) % chunk == 0 or i + 1 == steps:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append((ev, (i + 1)))
    torch.cuda.synchronize()
    if barrier:
        barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per = []
    prev_ev, prev_i = marks[0], 0
    for ev, i in marks[1:]:
        per.append(prev_ev.elapsed_time(ev) / (i - prev_i))
        prev_ev, prev_i = ev, i
    per.sort()
    chunks = {'steps_per_chunk': chunk, 'n_chunks': len(per), 'median_ms_per_step': round(per[len(per) // 2], 4),
              'min_ms_per_step': round(per[0], 4), 'max_ms_per_step': round(per[-1], 4)}
    return elapsed, chunks, out


def extra_train(model_name, batch, h, w, device, steps, warmup, loss='ce'):
    """One more BASELINE config in the same run (N = 1): same Trainer path as the headline, different model (or the recipe's
    OHEM loss, scripts/train_fastscnn.py, instead of the plain cross-entropy)."""
    from torch_semantic_segmentation_amd import engine as E
    model, tssa = build_model(model_name)
    model.to(device)
    tssa.set_compute_dtype(model, torch.bfloat16)
    opt = E.FlatAdamW(model.parameters(), lr=1e-3, weight_decay=1e-5)
    loss_fn = tssa.OHEMLoss(ignore_index=255, numel_frac=0.1) if loss == 'ohem' else tssa.CrossEntropyLoss(ignore_index=255)
    tr = E.Trainer(model, opt, loss_fn, use_graph=True)
    x, y = synthetic(batch, h, w, 1234, device)
    tr.step_async(x, y)
    x, y = tr.static_batch(x, y)
    for _ in range(warmup):
        tr.step_async(x, y)
    elapsed, chunks, loss = timed_steps(lambda: tr.step_async(x, y), steps, device)
    ms = 1e3 * elapsed / steps
    a = algorithmic_step_bytes(model_name, batch, h, w, 2)
    res = {'workload': '%s train step, %d x 3 x %d x %d, bf16%s' % (model_name, batch, h, w, ', OHEM loss' if loss == 'ohem' else ''),
           'ms_per_step': round(ms, 3),
           'images_per_sec': round(batch * steps / elapsed, 2), 'steps': steps, 'chunks': chunks, 'final_loss': round(float(loss), 4)}
    if a:
        res['step_roofline'] = roofline_block(a, ms)
    del tr, opt, model
    torch.cuda.empty_cache()
    return res


def extra_eval(model_name, h, w, device, steps, warmup):
    """BASELINE config 5: eval-mode forward of one 3 x 2048 x 4096 image through the x8 head, HIP-graph replay."""
    from torch_semantic_segmentation_amd import engine as E
    model, tssa = build_model(model_name)
    model.to(device).eval()
    tssa.set_compute_dtype(model, torch.bfloat16)
    x, _ = synthetic(1, h, w, 1234, device)
    fwd = E.GraphedInference(model, frozen_weights=True)      # inference on fixed weights: their preparation is not part of a forward
    x = fwd.static_input(x)
    with torch.no_grad():
        for _ in range(max(warmup, 2)):
            fwd(x)
        elapsed, chunks, _ = timed_steps(lambda: fwd(x), steps, device)
    ms = 1e3 * elapsed / steps
    S_ref = S_REF[model_name]
    a = (S_ref * h * w / (8.0 * 1024 * 2048) + 19.0 * h * w) * 2
    res = {'workload': '%s eval-mode forward incl. x8 head, 1 x 3 x %d x %d, bf16' % (model_name, h, w),
           'ms_per_step': round(ms, 3), 'images_per_sec': round(steps / elapsed, 2), 'steps': steps, 'chunks': chunks,
           'step_roofline': roofline_block(a, ms)}
    if model_name == 'fastscnn_aspp':
        rf = mfma_roofline(model, x)
        if rf:
            res['roofline'] = rf
    del fwd, model
    torch.cuda.empty_cache()
    return res


def mfma_roofline(model, x):
    """The dominant MATRIX kernel of an eval forward (the dense 3x3 convolutions of an ASPP head: 1152 FLOP per output byte, the only
    MFMA-bound launches of the repository): algorithmic FLOPs per launch / average launch time, measured live with HIP events on
    the launch stream (in-library profiler) over un-captured forwards, against the dense bf16 MFMA peak."""
    from torch_semantic_segmentation_amd import _native as N
    from torch_semantic_segmentation_amd import engine as E
    prep = E.EvalPrep(model)
    with torch.no_grad():
        with prep:
            model(x)
        torch.cuda.synchronize()
        N.prof_reset()
        N.prof_enable(True)
        for _ in range(3):
            with prep:
                model(x)
        torch.cuda.synchronize()
        N.prof_enable(False)
    r = N.prof_table().get('conv3x3_fwd')
    if not r or not r['launches'] or r['flops'] <= 0:
        return None
    tf = r['flops'] / (r['ms'] * 1e-3) / 1e12
    return {'bound': 'mfma', 'kernel': r['symbol'], 'achieved': round(tf, 1), 'peak': MFMA_BF16_PEAK_TF,
            'unit': 'TFLOP/s', 'frac': round(tf / MFMA_BF16_PEAK_TF, 4), 'traffic': None, 'launches_per_forward': r['launches'] // 3,
            'avg_launch_us': round(1e3 * r['ms'] / r['launches'], 2), 'alg_flops_per_launch': round(r['flops'] / r['launches'])}


def roofline_block(alg_bytes, ms):
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    return {'alg_bytes_per_step': round(alg_bytes), 'achieved_GBps': round(gbs, 1), 'frac_of_8TBps': round(gbs / HBM_PEAK_GBS, 4),
            'frac_of_measured_copy': round(gbs / HBM_COPY_GBS, 4)}


def main_eval(args):
    """SURVEY config C5 / TSS/utils/benchmark.py: eval-mode forward (running BatchNorm statistics, no dropout, no grad)
    through the x8 decoder head, HIP-graph replay, input resident in HBM."""
    from torch_semantic_segmentation_amd import engine as E
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    device = torch.device('cuda', 0)
    model, tssa = build_model(args.model)
    model.to(device).eval()
    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    tssa.set_compute_dtype(model, dtype)
    x, _ = synthetic(args.batch, args.height, args.width, 1234, device)
    fwd = E.GraphedInference(model, frozen_weights=True) if args.graph != 'off' else model
    if args.graph != 'off':
        x = fwd.static_input(x)      # the input is resident in the buffer the captured forward reads
    with torch.no_grad():
        for _ in range(max(args.warmup, 1)):
            out = fwd(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = fwd(x)
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    esz = 2 if dtype == torch.bfloat16 else 4
    ms = 1e3 * elapsed / args.steps
    S_ref = S_REF.get(args.model)
    res = {'metric': 'images/sec (eval forward) %s %dx%d bs=%d' % (args.model, args.height, args.width, args.batch),
           'value': round(args.batch * args.steps / elapsed, 2), 'unit': 'images/sec', 'n_gpus': 1, 'steps': args.steps,
           'warmup': args.warmup, 'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
           'dtype': args.dtype, 'data': 'synthetic',
           'config': {'workload': '%s eval-mode forward incl. x8 bilinear head, %d x 3 x %d x %d, 19 classes, random-init weights'
                                  % (args.model, args.batch, args.height, args.width), 'hip_graph': args.graph != 'off',
                      'logits': list(out.shape)}}
    if S_ref:   # A = (sum over blocks of in + out elements + full-resolution logits) * b
        a = (S_ref * args.batch * args.height * args.width / (8.0 * 1024 * 2048) + args.batch * 19.0 * args.height * args.width) * esz
        res['step_roofline'] = roofline_block(a, ms)
    if args.model == 'fastscnn_aspp' and dtype == torch.bfloat16:
        rf = mfma_roofline(model, x)
        if rf:
            res['roofline'] = rf
    print(json.dumps(res))


def main():
    args = parse()
    if args.mode == 'eval':
        return main_eval(args)
    from torch_semantic_segmentation_amd import engine as E
    from torch_semantic_segmentation_amd import _native as N
    # TSS_BENCH_REHEARSE=1: multi-rank dry run on a box with fewer GPUs than ranks (gloo collectives, ranks share the
    # visible devices round-robin) -- checks the N > 1 control flow, says nothing about speed
    rehearse = os.environ.get('TSS_BENCH_REHEARSE') == '1'
    if rehearse:
        ndev = max(torch.cuda.device_count(), 1)
        world, rank, local_rank = E.setup_distributed(enable=True, local_rank=int(os.environ.get('LOCAL_RANK', '0')) % ndev,
                                                      backend='gloo')
    else:
        world, rank, local_rank = E.setup_distributed(enable=True)
    if world != args.gpus and rank == 0:
        print('warning: --gpus %d but WORLD_SIZE %d' % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)

    model, tssa = build_model(args.model)
    model.to(device)
    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    tssa.set_compute_dtype(model, dtype)
    if args.syncbn:
        tssa.convert_syncbn_model(model)
    opt = E.FlatAdamW(model.parameters(), lr=1e-3, weight_decay=1e-5)
    loss_fn = tssa.CrossEntropyLoss(ignore_index=255)
    x, y = synthetic(args.batch, args.height, args.width, 1234 + rank, device)

    def make_trainer(use_graph):
        return E.Trainer(model, opt, loss_fn, device=None, use_graph=use_graph, world_size=world,
                         fuse_head_loss=args.fuse_head != 'off')

    graph_used = args.graph != 'off'
    trainer = make_trainer(graph_used)
    try:
        loss = trainer.step_async(x, y)
        torch.cuda.synchronize()
    except Exception as exc:  # graph capture unsupported -> eager launches (reported in config)
        if args.graph == 'on' or not graph_used:
            raise
        print('graph capture failed (%s: %s); falling back to eager launches' % (type(exc).__name__, exc), file=sys.stderr)
        graph_used = False
        trainer = make_trainer(False)
        loss = trainer.step_async(x, y)
        torch.cuda.synchronize()

    if graph_used:
        # the synthetic batch lives in the buffers the captured step reads (where a loader's H2D copy would put it):
        # inputs are resident in HBM when the timed region starts, no device-to-device staging copy inside it
        x, y = trainer.static_batch(x, y)
    for _ in range(max(args.warmup - 1, 0)):
        loss = trainer.step_async(x, y)
    barrier = dist.barrier if dist.is_initialized() else None
    elapsed, chunks, loss = timed_steps(lambda: trainer.step_async(x, y), args.steps, device, barrier=barrier)
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    final_loss = float(loss)
    if final_loss != final_loss or abs(final_loss) == float('inf'):
        raise SystemExit('bench.py: the training loss is not finite after %d steps -- the timed steps are not a valid workload' % args.steps)

    host_batch = None
    if args.host_batch:          # all ranks: the steps contain the gradient all-reduce
        # PCIe-inclusive rates (never the headline value): the batch crosses PCIe every step from pinned host memory,
        # (a) un-overlapped on the compute stream, (b) through engine.HostBatchPipeline as the reference's float32 / int64
        # tensors, (c) through the pipeline as uint8 pixels + uint8 labels decoded on the device
        n_hb = max(10, min(args.steps, 50))
        hx, hy = x.cpu().pin_memory(), y.cpu().pin_memory()
        host_batch = {'steps': n_hb, 'resident_ms_per_step': round(1e3 * elapsed / args.steps, 3)}
        for _ in range(2):
            trainer.step_async(hx, hy)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_hb):
            trainer.step_async(hx, hy)       # H2D straight into the captured step's buffers, same stream: no overlap
        torch.cuda.synchronize()
        host_batch['unoverlapped_ms_per_step'] = round(1e3 * (time.perf_counter() - t1) / n_hb, 3)
        ux = (hx * 40 + 128).clamp(0, 255).to(torch.uint8).pin_memory()
        uy = hy.to(torch.uint8).pin_memory()
        for wire, (bx, by), kw in (('f32', (hx, hy), {}), ('u8', (ux, uy), {'mean': (0.485, 0.456, 0.406), 'std': (0.229, 0.224, 0.225)})):
            pipe = E.HostBatchPipeline(trainer, hx, hy, wire=wire, **kw)
            pipe.put(bx, by)
            for _ in range(3):
                pipe.put(bx, by)
                pipe.step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n_hb):
                pipe.put(bx, by)
                pipe.step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / n_hb
            pipe.step()
            torch.cuda.synchronize()
            pipe.close()                    # its captured steps and input slots leave the trainer with it
            mb = (bx.numel() * bx.element_size() + by.numel() * by.element_size()) / 1e6
            host_batch['pipelined_%s' % wire] = {'ms_per_step': round(1e3 * dt, 3), 'images_per_sec': round(args.batch / dt, 1),
                                                 'wire_MB_per_step': round(mb, 1),
                                                 'vs_resident': round(dt / (elapsed / args.steps), 4)}
        x.copy_(hx, non_blocking=True)      # the roofline pass below runs on the original batch again
        y.copy_(hy, non_blocking=True)

    roofline = None
    breakdown = None
    if not args.no_roofline:     # every rank takes part (the un-captured steps all-reduce like the timed ones); rank 0 reports
        from torch_semantic_segmentation_amd import ops as _ops
        _ops.overlap_wgrad = False      # one kernel at a time, so that an event pair times exactly one kernel
        eager = make_trainer(False)
        eager.step_async(x, y)
        torch.cuda.synchronize()
        N.prof_reset()
        N.prof_enable(True)
        nprof = 3
        for _ in range(nprof):
            eager.step_async(x, y)
        torch.cuda.synchronize()
        N.prof_enable(False)
        table = N.prof_table()
        by_symbol = {}
        for op, r in table.items():
            s = by_symbol.setdefault(r['symbol'], dict(launches=0, ms=0.0, bytes=0.0, flops=0.0, ops=[]))
            s['launches'] += r['launches']; s['ms'] += r['ms']; s['bytes'] += r['bytes']; s['flops'] += r['flops']
            s['ops'].append(op)
        total_ms = sum(s['ms'] for s in by_symbol.values())
        sym, top = max(by_symbol.items(), key=lambda kv: kv[1]['ms'])
        gbs = top['bytes'] / (top['ms'] * 1e-3) / 1e9
        traffic, traffic_source = pmc_traffic(sym, args)
        roofline = {'bound': 'hbm', 'kernel': sym, 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(gbs / HBM_PEAK_GBS, 4), 'frac_of_measured_copy': round(gbs / HBM_COPY_GBS, 4),
                    'traffic': traffic, 'traffic_source': traffic_source,
                    'launches_per_step': top['launches'] // nprof,
                    'avg_launch_us': round(1e3 * top['ms'] / top['launches'], 2),
                    'alg_bytes_per_launch': round(top['bytes'] / top['launches']),
                    'share_of_kernel_time': round(top['ms'] / total_ms, 3),
                    'tflops': round(top['flops'] / (top['ms'] * 1e-3) / 1e12, 2)}
        breakdown = {k: {'ms_per_step': round(v['ms'] / nprof, 3), 'launches': v['launches'] // nprof,
                         'GBps': round(v['bytes'] / max(v['ms'], 1e-9) / 1e6, 1)}
                     for k, v in sorted(by_symbol.items(), key=lambda kv: -kv[1]['ms'])}
        if os.environ.get('TSS_BENCH_LAUNCHES'):
            recs = N.prof_records()
            per_step = len(recs) // nprof
            for i, (op, ms_i, by) in enumerate(recs[:per_step]):
                print('%4d %-24s %9.1f us %9.1f MB %8.1f GB/s' % (i, op, ms_i * 1e3, by / 1e6, by / max(ms_i, 1e-9) / 1e6),
                      file=sys.stderr)
        if os.environ.get('TSS_BENCH_OPS'):
            for op, r in sorted(table.items(), key=lambda kv: -kv[1]['ms']):
                print('%-24s %-28s launches/step %3d  ms/step %7.3f  GB/s %7.1f  TFLOP/s %6.2f' % (
                    op, r['symbol'], r['launches'] // nprof, r['ms'] / nprof, r['bytes'] / max(r['ms'], 1e-9) / 1e6,
                    r['flops'] / max(r['ms'], 1e-9) / 1e9), file=sys.stderr)

    if rank == 0:
        esz = 2 if dtype == torch.bfloat16 else 4
        ms = 1e3 * elapsed / args.steps
        value = world * args.batch * args.steps / elapsed
        a_step = algorithmic_step_bytes(args.model, args.batch, args.height, args.width, esz)
        out = {
            'metric': 'images/sec (train step) %s %dx%d bs=%d' % (args.model, args.height, args.width, args.batch),
            'value': round(value, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16' if dtype == torch.bfloat16 else 'f32', 'data': 'synthetic',
            'config': {'workload': '%s train step (zero_grad+fwd+CE(ignore 255)+bwd+AdamW), %d x 3 x %d x %d per GPU, '
                                   '19 classes, random-init weights' % (args.model, args.batch, args.height, args.width),
                       'global_batch': world * args.batch, 'parallelism': 'dp%d' % world,
                       'hip_graph': graph_used, 'final_loss': round(final_loss, 4), 'syncbn': bool(args.syncbn)},
            'timing': chunks,
        }
        if a_step:
            out['step_roofline'] = roofline_block(a_step, ms)
        if host_batch:
            out['host_batch'] = host_batch
        if roofline:
            out['roofline'] = roofline
            out['kernel_breakdown'] = breakdown
        default_workload = (args.model, args.batch, args.height, args.width, args.dtype) == ('fastscnn', 8, 1024, 2048, 'bf16')
        if world == 1 and default_workload and not args.no_extras:
            # BASELINE configs 3 and 5 in the same driver-run line (never the headline value)
            del trainer
            torch.cuda.empty_cache()
            n_extra = max(50, min(args.steps, 100))
            out['extra'] = {'contextnet14': extra_train('contextnet14', 8, 1024, 2048, device, n_extra, 5),
                            'fastscnn_ohem': extra_train('fastscnn', 8, 1024, 2048, device, n_extra, 5, loss='ohem'),
                            'eval_c5': extra_eval('fastscnn', 2048, 4096, device, n_extra, 5),
                            'eval_c5_aspp': extra_eval('fastscnn_aspp', 2048, 4096, device, n_extra, 5)}
        if args.stock:
            out['stock_pytorch_rocm_images_per_sec'] = round(stock_gpu(args.model, args.batch, args.height, args.width, device), 2)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.model, args.cpu_seconds)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
