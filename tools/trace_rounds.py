"""Per-kernel totals and the first rounds of the last step from a rocprofv3 --kernel-trace CSV (diagnostic)."""
import csv, re, sys, collections, glob
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_init' in r['Kernel_Name']]
seq = rows[idx[-1]:]
tot = collections.OrderedDict()
first = []
for r in seq:
    m = re.search(r'tk::(k_\w+)(<\w+>)?', r['Kernel_Name'])
    if not m:
        continue
    nm = m.group(1) + (m.group(2) or '')
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    t = tot.setdefault(nm, [0.0, 0]); t[0] += d; t[1] += 1
    if len(first) < int(sys.argv[2]) if len(sys.argv) > 2 else 60:
        first.append((nm, round(d), int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['LDS_Block_Size'], r['VGPR_Count']))
for k, v in tot.items():
    print(f"{k:16s} {v[0] / 1e3:9.2f} ms  {v[1]:5d} launches")
print('step wall us', (int(seq[-1]['End_Timestamp']) - int(seq[0]['Start_Timestamp'])) / 1e3)
for x in first:
    print(x)
