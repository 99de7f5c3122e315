#!/usr/bin/env python
"""MFMA utilisation per kernel from one rocprofv3 counter pass (--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE).
usage: pmc_mfma.py <counter_dir>
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): MFMA_BUSY is summed over the 1024 SIMDs and counts cycles
(16 per v_mfma_f32_16x16x32_bf16, MI355X_MICROARCH.md); rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (checked:
value / 8 / 2.4 GHz = the kernel's duration), so one XCD's active cycles are value / 8.  Every hot-path layer is
HBM-bound (arithmetic intensity 55-64 FLOP/B against a machine balance of ~310), so single-digit MFMA utilisation is the
expected reading; the number is reported because the task statement asks for it."""
import collections
import csv
import glob
import re
import sys

SIMDS, XCDS = 256 * 4, 8
val = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')
        n = re.sub(r'\(.*', '', n)
        val[n][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            cnt[n] += 1
print('%-58s %8s %14s %14s %9s' % ('kernel', 'launches', 'mfma_busy/launch', 'gui_active/launch', 'MfmaUtil'))
rows = []
for n, c in val.items():
    if c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) <= 0 or cnt[n] == 0:
        continue
    busy, act = c['SQ_VALU_MFMA_BUSY_CYCLES'] / cnt[n], c['GRBM_GUI_ACTIVE'] / cnt[n]
    rows.append((busy / (act / XCDS * SIMDS), n, cnt[n], busy, act))
for u, n, k, busy, act in sorted(rows, reverse=True):
    print('%-58s %8d %14.0f %14.0f %8.2f%%' % (n[:58], k, busy, act, 100 * u))
