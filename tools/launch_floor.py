import torch, time
a = torch.zeros(64, device='cuda')
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): a.add_(1)
torch.cuda.current_stream().wait_stream(s)
for n in (300,):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): a.add_(1)
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print('graph of %d dependent tiny kernels: %.1f us per replay, %.2f us per kernel' % (n, dt * 1e6, dt * 1e6 / n))
# eager stream launches
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(2000): a.add_(1)
torch.cuda.synchronize(); dt = time.perf_counter() - t
print('eager: %.2f us per kernel' % (dt * 1e6 / 2000))
