#!/usr/bin/env python
"""Where the waves of each kernel spend their cycles (rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT, one counter-only pass over un-captured steps):
WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY (issuing) ~ WAVE_CYCLES, per
MI355X_MICROARCH.md.  usage: pmc_wavestate.py <rocprofv3 output dir>"""
import collections
import csv
import glob
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(.*', '', r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', ''))[:60]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'SQ_WAVE_CYCLES':
            cnt[k] += 1
rows = []
for k, d in acc.items():
    wc = d.get('SQ_WAVE_CYCLES', 0.0)
    if wc <= 0:
        continue
    rows.append((wc, k, cnt[k], d.get('SQ_WAIT_ANY', 0) / wc, d.get('SQ_WAIT_INST_ANY', 0) / wc, d.get('SQ_ACTIVE_INST_ANY', 0) / wc,
                 d.get('SQ_ACTIVE_INST_VALU', 0) / wc, d.get('SQ_INSTS_VALU', 0) / max(cnt[k], 1),
                 d.get('SQ_LDS_BANK_CONFLICT', 0) / max(d.get('SQ_INSTS_LDS', 0), 1)))
rows.sort(reverse=True)
print('%-60s %5s %7s %7s %7s %7s %12s %9s' % ('kernel (sorted by wave cycles)', 'n', 'wait', 'stall', 'issue', 'valu', 'VALU/launch', 'bankc/LDS'))
for wc, k, n, w, s_, a, v, iv, bc in rows[:32]:
    print('%-60s %5d %6.1f%% %6.1f%% %6.1f%% %6.1f%% %12.3g %9.2f' % (k, n, 100 * w, 100 * s_, 100 * a, 100 * v, iv, bc))
