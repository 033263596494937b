"""Timeline of the LAST LM iteration in a rocprofv3 kernel trace: lm_timeline.py kernel_trace.csv
An iteration starts at a J^T J pass (k_run_pass<...JtjOp...>).  Prints consecutive launches of one kernel as one line:
count, first start (us from the iteration's start), sum of durations, sum of the idle gaps before each."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows))
starts = [k for k, e in enumerate(ev) if 'JtjOp' in e[2]]
a, b = starts[-2], starts[-1]
it = ev[a:b]
t0 = it[0][0]
print(f'iteration of {len(it)} launches, {(ev[b][0] - t0) / 1e3:.1f} us')
groups = []
prev_end = t0
for s, e, n in it:
    n = n.split('(')[0][:44]
    gap = max(s - prev_end, 0)
    if groups and groups[-1][0] == n:
        g = groups[-1]
        g[1] += 1; g[3] += e - s; g[4] += gap
    else:
        groups.append([n, 1, s - t0, e - s, gap])
    prev_end = max(prev_end, e)
for n, c, s, d, g in groups:
    print(f'{s / 1e3:9.1f} us  x{c:3d}  busy {d / 1e3:8.1f}  idle before {g / 1e3:7.1f}  {n}')
tot = {}
for s, e, n in it:
    n = n.split('(')[0][:44]
    tot[n] = (tot.get(n, (0, 0))[0] + e - s, tot.get(n, (0, 0))[1] + 1)
print('--- totals')
for n, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f'{d / 1e3:9.1f} us  x{c:3d}  {n}')
