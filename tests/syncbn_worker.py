"""Worker of tests/test_gpu_models.py::test_syncbn_two_ranks_equal_one_big_batch (launched by torch.distributed.run with
2 ranks, gloo collectives, both ranks on cuda:0): a small stack with cross-replica BatchNorm on this rank's half of a
fixed global batch; rank 0 saves its outputs, the rank-summed parameter gradients and the running statistics."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build():
    import torch_semantic_segmentation_amd as tssa
    import importlib
    F = importlib.import_module('torch_semantic_segmentation_amd.models.fastscnn')   # `models.fastscnn` is also a function
    torch.manual_seed(0)
    model = torch.nn.Sequential(F.Conv2dBlock(16, 32, kernel_size=1), F.BottleneckBlock(32, 32, expansion=6),
                                F.DSConv2dBlock(32, 48, kernel_size=3, padding=1, stride=2))
    for p in model.parameters():
        if p.dim() == 1:
            p.data.uniform_(0.5, 1.5)
    tssa.set_compute_dtype(model, torch.float32)
    return model


def batch():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 16, 16, 32, generator=g)
    cot = torch.randn(4, 48, 8, 16, generator=g)
    return x, cot


def run(model, x, cot):
    model.train()
    out = model(x)
    (out.float() * cot).sum().backward()
    return out.detach().float().contiguous()


if __name__ == '__main__':
    import torch_semantic_segmentation_amd as tssa
    dist.init_process_group('gloo', init_method='env://')
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    model = build().to(dev)
    tssa.convert_syncbn_model(model)
    x, cot = batch()
    lo, hi = rank * 4 // world, (rank + 1) * 4 // world
    out = run(model, x[lo:hi].to(dev), cot[lo:hi].to(dev))
    grads = {}
    for n, p in model.named_parameters():
        g = p.grad.detach().clone()
        dist.all_reduce(g)                       # sum over the ranks == gradient of the global sum-loss
        grads[n] = g.cpu()
    from torch_semantic_segmentation_amd import ops
    exs = [e for e in ops._XCHG.values()]
    ipc = bool(exs) and all(e is not None for e in exs)
    xerr = max([e.error() for e in exs if e is not None] or [0])
    if rank == 0:
        torch.save({'out': out.cpu(), 'grads': grads, 'ipc': ipc, 'xerr': xerr,
                    'buffers': {n: b.detach().cpu() for n, b in model.named_buffers()}}, sys.argv[1])
    dist.barrier()
