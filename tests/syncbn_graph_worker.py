"""Worker of tests/test_gpu_models.py::test_syncbn_step_is_captured_in_a_hip_graph_with_rccl: one rank, RCCL backend,
TSS_SYNCBN_FORCE=1.  Trains FastSCNN for a few steps twice -- convert_syncbn_model + Trainer(use_graph=True), and plain
BatchNorm un-captured -- and saves both loss curves and whether the SyncBatchNorm step really ran as a HIP graph."""
import os
import sys
import warnings

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    import torch_semantic_segmentation_amd as tssa
    from torch_semantic_segmentation_amd import engine as E
    from torch_semantic_segmentation_amd import ops
    from torch_semantic_segmentation_amd.models.fastscnn import fastscnn
    from oracle.recipe import synthetic_batch
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method='env://')
    dev = torch.device('cuda', 0)
    x, y = synthetic_batch(2, 64, 128)
    x, y = x.to(dev), y.to(dev)
    curves, captured, warning, n_sync = {}, False, '', 0
    for mode in ('sync', 'local'):
        torch.manual_seed(0)
        m = fastscnn(3, 19).to(dev)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        if mode == 'sync':
            tssa.convert_syncbn_model(m)
            n_sync = sum(ops._sync_group(b, any_mode=True) is not None for b in m.modules()
                         if isinstance(b, torch.nn.modules.batchnorm._BatchNorm))
        opt = E.FlatAdamW(m.parameters(), lr=1e-3, weight_decay=1e-5)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter('always')
            tr = E.Trainer(m, opt, tssa.CrossEntropyLoss(ignore_index=255), use_graph=(mode == 'sync'))
            curves[mode] = [tr.step_async(x, y).item() for _ in range(4)]
        if mode == 'sync':
            captured = tr._graph is not None and tr.use_graph
            warning = '; '.join(str(w.message) for w in caught)
    # the gradient all-reduce captured INSIDE the step (Trainer(graph_allreduce=True)): the replayed step must give
    # the trajectory of the eager-collective trainer, with the collective gone from the host side of the step
    ar = {}
    for mode in ('graph_allreduce', 'eager_allreduce'):
        torch.manual_seed(0)
        m = fastscnn(3, 19).to(dev)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        tssa.set_compute_dtype(m, torch.bfloat16)      # the benchmarked kernels: no atomics, so the two trajectories must be bit-identical
        opt = E.FlatAdamW(m.parameters(), lr=1e-3, weight_decay=1e-5)
        tr = E.Trainer(m, opt, tssa.CrossEntropyLoss(ignore_index=255), use_graph=True, graph_allreduce=(mode == 'graph_allreduce'))
        ar[mode] = [tr.step_async(x, y).item() for _ in range(4)]
        ar[mode + '_captured'] = bool(tr.use_graph and tr._reduce_captured)
    torch.save({'losses_sync': curves['sync'], 'losses_local': curves['local'], 'captured': captured, 'warning': warning,
                'sync_layers': n_sync, 'allreduce': ar}, sys.argv[1])
    dist.barrier()
    dist.destroy_process_group()
