#!/usr/bin/env python
"""Probe (GPU box): does a replayed HIP graph run two forked branches concurrently, and does it depend on which branch was captured
first / on what the fork node's first successor is?  Each branch is a chain of N spin kernels (torch.cuda._sleep); a replay that
overlaps them takes about one chain, a serial one two.  Variants:
  side_first    fork -> [side chain captured first] + [main chain]      (ContextNet's forward as of round 3)
  main_first    fork -> [main chain captured first] + [side chain]
  dummy_first   fork -> a one-kernel third branch captured FIRST, then side chain, then main chain
  two_sides     fork -> both chains on side streams, nothing on the capture stream
"""
import sys
import torch

dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
CYC = int(sys.argv[2]) if len(sys.argv) > 2 else 40000


def chain():
    for _ in range(N):
        torch.cuda._sleep(CYC)


def build(variant):
    x = torch.zeros(1024, device=dev)
    d = torch.zeros(1024, device=dev)
    s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        x.add_(1)                                        # the fork node
        if variant == 'serial':
            chain(); chain()
        elif variant == 'side_first':
            s1.wait_stream(main)
            with torch.cuda.stream(s1):
                chain()
            chain()
            main.wait_stream(s1)
        elif variant == 'main_first':
            s1.wait_stream(main)
            chain()
            with torch.cuda.stream(s1):
                chain()
            main.wait_stream(s1)
        elif variant == 'dummy_first':
            s3.wait_stream(main)
            with torch.cuda.stream(s3):
                d.add_(1)
            s1.wait_stream(main)
            with torch.cuda.stream(s1):
                chain()
            chain()
            main.wait_stream(s1)
            main.wait_stream(s3)
        elif variant == 'two_sides':
            s1.wait_stream(main)
            s2.wait_stream(main)
            with torch.cuda.stream(s1):
                chain()
            with torch.cuda.stream(s2):
                chain()
            main.wait_stream(s1)
            main.wait_stream(s2)
        x.add_(1)                                        # the join node
    return g, (x, d, s1, s2, s3)


for variant in ('serial', 'side_first', 'main_first', 'dummy_first', 'two_sides'):
    g, keep = build(variant)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    print('%-12s %8.1f us per replay (2 chains of %d spin kernels)' % (variant, a.elapsed_time(b) / 10 * 1e3, N))
