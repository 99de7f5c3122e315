#!/bin/bash
# usage: tools/pmc_ab.sh <tag> <micro_one args...>   (TSS_OPT in the environment selects the variant)
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $out/a --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/micro_one.py "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA -d $out/b --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/micro_one.py "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $out/t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/micro_one.py "$@" > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for sub in 'ab':
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob('$out/%s/**/*counter_collection.csv' % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:60]
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); 
            cnt[(k, r['Counter_Name'])] += 1
    for k, d in agg.items():
        if any(t in k for t in ('pw', 'wg', 'convgemm', 'dw_', 'upsample', 'ce_')):
            print('$tag', k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
for f in glob.glob('$out/t/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(t in r['Name'] for t in ('pw', 'wg', 'convgemm', 'dw_', 'upsample', 'ce_')):
            print('$tag', r['Name'][:60], 'avg_ns', r['AverageNs'])
PY
