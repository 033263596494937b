"""Per-iteration kernel durations and gaps over a whole bench run from a rocprofv3 kernel trace:
warmup_trace.py <dir> -> windows of 50 iterations: mean duration of the pass, of the camera step, of the gap
between the two and of the gap before the next pass (us)."""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
its = []
for a, b in zip(seq, seq[1:]):
    if 'k_run_pass' in a[0] and 'k_camera_step_quad' in b[0]:
        its.append((a[1], a[2], b[1], b[2]))
print(len(its), 'iterations (pass followed by camera step)')
its = np.array(its, dtype=np.float64) / 1e3
for w in range(0, len(its) - 1, 50):
    blk = its[w:w + 50]
    nxt = its[w + 1:w + 51, 0][:len(blk)]
    blk = blk[:len(nxt)]
    print(f'iterations {w:4d}..{w + len(blk) - 1:4d}: pass {np.mean(blk[:, 1] - blk[:, 0]):6.2f}  camera step '
          f'{np.mean(blk[:, 3] - blk[:, 2]):6.2f}  gap pass->step {np.mean(blk[:, 2] - blk[:, 1]):5.2f}  gap step->pass '
          f'{np.mean(nxt - blk[:, 3]):5.2f}  period {np.mean(nxt - blk[:, 0]):6.2f}')
