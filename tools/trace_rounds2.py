"""Per-round kernel times and idle gaps of the last step from a rocprofv3 --kernel-trace CSV (diagnostic)."""
import csv, re, sys, glob
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_init' in r['Kernel_Name']]
seq = rows[idx[-1]:]
rounds = []
cur = None
prev_end = int(seq[0]['Start_Timestamp'])
for r in seq:
    m = re.search(r'tk::(k_\w+)(<[\w, ]+>)?', r['Kernel_Name'])
    nm = (m.group(1) + (m.group(2) or '')) if m else 'other'
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if nm == 'k_loop' or cur is None:
        cur = {'t0': st, 'k': {}, 'gap': 0.0, 'grid': {}}
        rounds.append(cur)
    cur['k'][nm] = cur['k'].get(nm, 0.0) + (en - st) / 1e3
    cur['grid'][nm] = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])
    cur['gap'] += max(0, st - prev_end) / 1e3
    prev_end = max(prev_end, en)
    cur['t1'] = en
tot_gap = 0
for i, c in enumerate(rounds):
    tot_gap += c['gap']
    print(i, 'wall %.0f us gap %.0f |' % ((c['t1'] - c['t0']) / 1e3, c['gap']), ' '.join('%s %.0f(%d)' % (k, v, c['grid'][k]) for k, v in c['k'].items()))
print('step wall ms', (int(seq[-1]['End_Timestamp']) - int(seq[0]['Start_Timestamp'])) / 1e6, 'gaps ms', tot_gap / 1e3)
