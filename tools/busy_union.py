"""From a rocprofv3 --kernel-trace CSV: how much of the timed region the GPU ran nothing, only small launches (fewer workgroups
than 4 per CU), or at least one large launch (diagnostic)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f))]
ev = []
for r in rows:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    wgs = (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))) * max(1, int(r['Grid_Size_Y']) // max(1, int(r['Workgroup_Size_Y'])))
    big = wgs >= 1024
    ev.append((st, 1, big)); ev.append((en, -1, big))
ev.sort()
t0 = ev[len(ev) // 4][0]; t1 = ev[-len(ev) // 8][0]       # the middle of the run (skips set-up and the exclusive step)
nb = ns = 0; last = None
acc = {'idle': 0, 'small only': 0, 'large': 0}; conc = 0.0
for t, d, big in ev:
    if last is not None and t > t0 and last < t1:
        a, b = max(last, t0), min(t, t1)
        if b > a:
            k = 'large' if nb else ('small only' if ns else 'idle')
            acc[k] += b - a; conc += (nb + ns) * (b - a)
    if big: nb += d
    else: ns += d
    last = t
tot = sum(acc.values())
print({k: round(v / tot, 3) for k, v in acc.items()}, 'mean launches in flight %.2f' % (conc / tot), 'window %.1f ms' % (tot / 1e6))
