"""First few k_aln / k_job / k_loop launch durations from a rocprofv3 --kernel-trace CSV (diagnostic)."""
import csv, sys, glob, re
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
seen = {}
for r in rows:
    m = re.search(r'tk::(k_\w+)', r['Kernel_Name'])
    if not m: continue
    k = m.group(1)
    if k in ('k_aln', 'k_job', 'k_loop') and seen.get(k, 0) < 3:
        seen[k] = seen.get(k, 0) + 1
        print(k, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 'us', int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), 'waves')
