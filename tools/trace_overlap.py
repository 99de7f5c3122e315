#!/usr/bin/env python
"""Concurrency of a step timeline (gpurun_out/timeline.txt of tools/trace_step.sh): time covered by 0 / 1 / 2+ kernels, and the
first and last moment two kernels ran together -- for the two-stream regions of ContextNet.  usage: trace_overlap.py <timeline.txt>"""
import sys

rows = []
for line in open(sys.argv[1]):
    p = line.split(None, 2)
    rows.append((float(p[0]), float(p[1]), p[2].strip()))
ev = []
for s, d, _ in rows:
    ev.append((s, 1))
    ev.append((s + d, -1))
ev.sort()
cov = {0: 0.0, 1: 0.0, 2: 0.0}
cur, last, first2, last2 = 0, 0.0, None, None
for t, x in ev:
    cov[min(cur, 2)] += t - last
    if cur >= 2:
        first2 = last if first2 is None else first2
        last2 = t
    last = t
    cur += x
print('idle %.1f us, one kernel %.1f us, two or more %.1f us; concurrency between %s and %s us of %.1f' % (
    cov[0], cov[1], cov[2], first2, last2, rows[-1][0] + rows[-1][1]))
