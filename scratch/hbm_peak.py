"""What the HBM of the box delivers to simple streaming kernels (torch ops): copy, triad, read-only sum, fill."""
import torch
dev = 'cuda:0'
n = 1 << 27                                   # 1 GiB per fp64 tensor
a = torch.rand(n, dtype=torch.float64, device=dev); b = torch.rand_like(a); c = torch.empty_like(a)
def timed(fn, bytes_moved, label, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f'{label:28s} {ms*1e3:9.1f} us  {bytes_moved / ms / 1e6:8.1f} GB/s', flush=True)
timed(lambda: c.copy_(a), 16 * n, 'copy (1 read + 1 write)')
timed(lambda: torch.add(a, b, alpha=1.5, out=c), 24 * n, 'triad (2 reads + 1 write)')
timed(lambda: a.sum(), 8 * n, 'sum (1 read)')
timed(lambda: c.fill_(1.0), 8 * n, 'fill (1 write)')
m = 12_500_000                                 # the size of one stream of the pass at cfg3 (100 MB)
x = torch.rand(m, dtype=torch.float64, device=dev); y = torch.empty_like(x)
timed(lambda: y.copy_(x), 16 * m, 'copy of 100 MB (cache resident)', reps=200)
