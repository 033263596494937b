"""Idle time between the kernels of LM iterations, from a rocprofv3 kernel trace: lm_gaps.py kernel_trace.csv
Prints, for the second half of the trace, the busy and idle time and the largest gaps with the kernels on either side."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:50]) for r in rows))
ev = ev[len(ev) // 2:]
busy = sum(e - s for s, e, _ in ev)
span = ev[-1][1] - ev[0][0]
gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
    if s1 > e0:
        gaps.append((s1 - e0, n0, n1))
print(f'{len(ev)} kernels over {span / 1e6:.2f} ms: busy {busy / 1e6:.2f} ms, idle {sum(g for g, _, _ in gaps) / 1e6:.2f} ms in {len(gaps)} gaps')
by = {}
for g, n0, n1 in gaps:
    k = (n0, n1)
    by[k] = (by.get(k, (0, 0))[0] + g, by.get(k, (0, 0))[1] + 1)
for (n0, n1), (g, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f'{g / 1e3:9.1f} us in {c:4d} gaps ({g / c / 1e3:6.1f} each)  {n0}  ->  {n1}')
