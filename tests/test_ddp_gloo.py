"""CPU, 2 processes, gloo: the data-parallel contract of SURVEY.md section 8e --
averaged per-rank gradients == gradients of the mean loss over the concatenated batch (identical BN statistics),
one all-reduce over the flat gradient buffer, identical parameters on every rank afterwards."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


class Net(nn.Module):
    """conv stack with *frozen* BatchNorm, so per-shard and whole-batch statistics coincide."""

    def __init__(self):
        super().__init__()
        self.body = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(),
                                  nn.Conv2d(8, 8, 3, padding=1, groups=8), nn.BatchNorm2d(8), nn.ReLU(),
                                  nn.Conv2d(8, 5, 1))

    def train(self, mode=True):
        super().train(mode)
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()
        return self

    def forward(self, x):
        return self.body(x)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from torch_semantic_segmentation_amd import engine as E
    w, r, _ = E.setup_distributed(enable=True, backend='gloo')
    assert (w, r) == (world, rank)
    torch.manual_seed(0)
    model = Net()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 3, 12, 12, generator=g)
    y = torch.randint(0, 5, (8, 12, 12), generator=g)
    idx = E.shard_batch(8, world, rank)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, weight_decay=1e-5)
    tr = E.Trainer(model, opt, nn.CrossEntropyLoss())          # world size derived from the live process group
    assert tr.world_size == world
    try:
        E.Trainer(model, opt, nn.CrossEntropyLoss(), world_size=world + 1)
        raise AssertionError('a world_size that disagrees with torch.distributed must be refused')
    except ValueError:
        pass
    tr.step_async(x[idx], y[idx])
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    grads = torch.cat([p.grad.flatten() for p in model.parameters()])
    # the flat-buffer collective used by FlatAdamW: ONE all_reduce, sum (the 1/world is folded into the optimizer)
    fopt = E.FlatAdamW(Net().parameters(), lr=1e-3)
    fopt.flat_grad.fill_(float(rank + 1))
    E.allreduce_mean_(fopt.flat_grad, world)
    if rank == 0:
        torch.save({'params': gathered, 'grads': grads, 'flat_sum': fopt.flat_grad[:4].clone(),
                    'alias': next(iter(fopt.param_groups[0]['params'])).grad.flatten()[:2].clone()}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_average_equals_single_process(tmp_path):
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    assert torch.equal(got['params'][0], got['params'][1])           # replicas stay in lockstep
    assert torch.equal(got['flat_sum'], torch.full((4,), 3.0)) and torch.equal(got['alias'], torch.full((2,), 3.0))
    # single process on the concatenated batch
    torch.manual_seed(0)
    model = Net()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 3, 12, 12, generator=g)
    y = torch.randint(0, 5, (8, 12, 12), generator=g)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, weight_decay=1e-5)
    opt.zero_grad()
    nn.CrossEntropyLoss()(model(x), y).backward()
    ref_grads = torch.cat([p.grad.flatten() for p in model.parameters()])
    opt.step()
    ref_params = torch.cat([p.detach().flatten() for p in model.parameters()])
    assert torch.allclose(got['grads'], ref_grads, rtol=1e-4, atol=1e-6)
    assert torch.allclose(got['params'][0], ref_params, rtol=1e-4, atol=1e-6)
