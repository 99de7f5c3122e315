#!/usr/bin/env python
"""Per-entry-point roofline table (markdown) from a bench.py JSON line: tools/roofline_table.py profiles/r01_bench_default.json
ms/step, launches/step, algorithmic GB/s (HIP-event time of the un-captured profiling pass) and the fraction of 8 TB/s."""
import json
import sys

d = json.load(open(sys.argv[1]))
kb = d['kernel_breakdown']
tot = sum(v['ms_per_step'] for v in kb.values())
print('| entry point (kernels) | ms / step | launches | algorithmic GB/s | frac of 8 TB/s |')
print('|---|---|---|---|---|')
for k, v in kb.items():
    if v['ms_per_step'] < 0.02:
        continue
    print('| `%s` | %.3f | %d | %.0f | %.1f %% |' % (k, v['ms_per_step'], v['launches'], v['GBps'], 100 * v['GBps'] / 8000.0))
print('\nsum of event-timed kernels: %.2f ms (un-captured pass: every launch timed on its stream)' % tot)
