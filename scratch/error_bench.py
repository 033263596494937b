"""ba_reprojection_error and problem set-up at cfg3 size: error_bench.py"""
import importlib, sys, os, time, numpy as np, torch, ctypes as C
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
ba = importlib.import_module('3d_reconstruction_amd')
eng = ba.engine
lib = eng._lib.load()
n = 10_000_000
pix = torch.randint(0, 1000, (n, 2), dtype=torch.int32, device='cuda:0')
model = pix.to(torch.float64) + 0.7
sums = torch.zeros(2, dtype=torch.float64, device='cuda:0')
for rep in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        eng.check(lib.ba_reprojection_error(n, eng.ptr(pix), eng.ptr(model), eng.ptr(sums), eng.stream_ptr()))
    b.record(); torch.cuda.synchronize()
print(f'ba_reprojection_error, {n} observations: {a.elapsed_time(b) / 20 * 1e3:.1f} us per call, sums {sums.tolist()}')
sc = ba.scenes.orbit_scene(N=1_000_000, K=2000, track_len=10, seed=0)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sh = eng.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], sc['K'], sc['W'], sc['H'], 'cuda:0')
    torch.cuda.synchronize(); print(f'Shard set-up {1e3 * (time.perf_counter() - t0):.1f} ms', flush=True)
    sh.close()
