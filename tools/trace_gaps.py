#!/usr/bin/env python
"""Per-step timeline summary from a rocprofv3 --kernel-trace CSV: for the last full step (delimited by adamw_kernel, or the kernel named in $TRACE_MARK),
wall time, sum of kernel durations, idle gaps and per-kernel totals.  usage: trace_gaps.py <kernel_trace.csv>"""
import collections
import csv
import os
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
mark = os.environ.get('TRACE_MARK', 'adamw_kernel')   # the kernel that ends a step (eval traces: the x8 head)
marks = [i for i, r in enumerate(rows) if mark in r[2]]
if len(marks) < 3:
    sys.exit('need at least 3 steps in the trace')
lo, hi = marks[-2] + 1, marks[-1] + 1   # one full step: after the previous AdamW up to and including this one
step = rows[lo:hi]
wall = step[-1][1] - step[0][0]
busy = sum(e - s for s, e, _ in step)
gap = 0
prev_end = step[0][0]
for s, e, _ in step:
    if s > prev_end:
        gap += s - prev_end
    prev_end = max(prev_end, e)
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in step:
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    n = re.sub(r'\(.*', '', n)
    agg[n][0] += e - s
    agg[n][1] += 1
if len(sys.argv) > 2:
    with open(sys.argv[2], 'w') as f:
        for s_, e_, n_ in step:
            f.write(f'{(s_ - step[0][0]) / 1e3:9.1f} {(e_ - s_) / 1e3:8.1f}  {n_[:110]}\n')
print(f'kernels {len(step)}  wall {wall / 1e3:.1f} us  sum(durations) {busy / 1e3:.1f} us  idle gaps {gap / 1e3:.1f} us')
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f'{t / 1e3:9.1f} us {c:4d}  avg {t / c / 1e3:7.1f}  {n[:90]}')
