"""Kernel sequence of the last iterations in a rocprofv3 kernel trace: name, duration, gap to the previous kernel (us)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tail = rows[-n:]
prev = int(rows[-n - 1]['End_Timestamp'])
for r in tail:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{r['Kernel_Name'][:70]:70s} dur {(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:7.1f}")
    prev = e
