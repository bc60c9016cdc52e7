"""Per-kernel sums of the counters of a rocprofv3 --pmc run (diagnostic): python tools/pmc_kernels.py <dir>"""
import csv, sys, glob, re, collections
f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f)):
    m = re.search(r'tk::(k_\w+)(<[\w, ]+>)?', r['Kernel_Name'])
    if not m: continue
    k = m.group(1) + (m.group(2) or '')
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in acc.items():
    print(k, ' '.join('%s=%.4g' % (c, v) for c, v in sorted(d.items())))
