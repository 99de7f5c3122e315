"""Average of every PMC counter of the launches whose kernel name contains a pattern (rocprofv3 --pmc ... --output-format csv).
usage: python tools/pmc_kernel.py <dir> <pattern>"""
import csv, glob, sys, collections
d, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r.get('Kernel_Name', ''):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print('%-40s launches %4d  mean %.4g' % (k, len(v), sum(v) / len(v)))
