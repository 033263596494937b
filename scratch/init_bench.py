"""Triangulation and resection at cfg3 size: time and bytes streamed."""
import importlib, sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
if os.environ.get('BA_LIB'):
    importlib.import_module('3d_reconstruction_amd._lib').LIB_PATH = os.path.abspath(os.environ['BA_LIB'])
ba = importlib.import_module('3d_reconstruction_amd')
sc = ba.scenes.orbit_scene(N=1_000_000, K=2000, track_len=10, seed=0)
N, K, W, H = sc['N_local'], sc['K'], sc['W'], sc['H']
dev = 'cuda:0'
oi = torch.from_numpy(sc['i']).to(dev); oj = torch.from_numpy(sc['j']).to(dev); uv = torch.from_numpy(sc['uv']).to(dev)
C, Th, X = (torch.from_numpy(sc[k]).to(dev) for k in ('C', 'Theta', 'X'))
nv = oi.numel()
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); out = fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts)), out
ms, Xd = timeit(lambda: ba.engine.triangulate(oi, oj, uv, N, C, Th, sc['phi_x'], W, H, False, dev))
print(f'triangulate (wrapper incl. index build): {ms:.3f} ms for {nv} observations, max |X - truth| = {float((Xd - X).abs().nan_to_num().max()):.3e}', flush=True)
ms, (Cd, Td) = timeit(lambda: ba.engine.resect(oi, oj, uv, K, X, sc['phi_x'], W, H, False, dev))
print(f'resect (wrapper incl. sort by camera): {ms:.3f} ms, max |C - truth| = {float((Cd - C).abs().nan_to_num().max()):.3e}', flush=True)
