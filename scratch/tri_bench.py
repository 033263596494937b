"""k_triangulate at cfg3 size on the caller's (random) point order and on the engine's order: tri_bench.py"""
import importlib, sys, os, numpy as np, torch, ctypes as C
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
if os.environ.get("BA_LIB"):
    importlib.import_module("3d_reconstruction_amd._lib").LIB_PATH = os.path.abspath(os.environ["BA_LIB"])
ba = importlib.import_module("3d_reconstruction_amd")
eng = ba.engine
lib = eng._lib.load()
sc = ba.scenes.orbit_scene(N=1_000_000, K=2000, track_len=10, seed=0)
N, K, W, H = sc['N_local'], sc['K'], sc['W'], sc['H']
dev = 'cuda:0'
def run(i, j, uv, tag):
    oi = torch.from_numpy(i).to(dev); oj = torch.from_numpy(j).to(dev); uvd = torch.from_numpy(uv).to(dev)
    Cd, Td = (torch.from_numpy(sc[k]).to(dev) for k in ('C', 'Theta'))
    row_ptr, counts = eng._ranges(oi, N)
    X = torch.empty((N, 3), dtype=torch.float64, device=dev)
    P = torch.empty(16 * K, dtype=torch.float64, device=dev)
    ts = []
    for rep in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.check(lib.ba_triangulate(N, K, eng.ptr(row_ptr), eng.ptr(oj), eng.ptr(uvd), eng.ptr(Cd), eng.ptr(Td), float(sc['phi_x']),
                                     float(W), float(H), 0, 2 * K, eng.ptr(P), eng.ptr(X), eng.stream_ptr()))
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    nv = i.size
    byts = 20 * nv + 8 * (N + 1) + 24 * N
    t = float(np.median(ts[2:]))
    print(f'{tag:28s} {t:8.1f} us  {byts / t / 1e6:6.2f} TB/s = {byts / t / 1e6 / 8 * 100:5.1f} % of 8 TB/s', flush=True)
    return X
run(sc['i'], sc['j'], sc['uv'], 'caller order (random)')
first = np.full(N, K); np.minimum.at(first, sc['i'], sc['j'])
perm = np.argsort(first, kind='stable'); inv = np.empty(N, np.int64); inv[perm] = np.arange(N)
ni = inv[sc['i']]; order = np.argsort(ni, kind='stable')
run(ni[order].astype(np.int32), sc['j'][order], sc['uv'][order], 'engine order (first camera)')
sh = eng.Shard(sc['i'], sc['j'], sc['uv'], N, K, W, H, dev)
sh.set_params(sc['X_0'], sc['C'], sc['Theta'], sc['phi_x'])
ts = []
for rep in range(8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); Xs = sh.triangulate(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
print(f'Shard.triangulate (incl. permutation back) {float(np.median(ts[2:])):8.1f} us, max |X - truth| {float((Xs - torch.from_numpy(sc["X"]).to(dev)).abs().nan_to_num().max()):.3e}')
cam = sh.cam; c = cam.data_ptr(); K24 = 24 * K
Xo = torch.empty((N, 3), dtype=torch.float64, device=dev); Pw = torch.empty(12 * K, dtype=torch.float64, device=dev)
ts = []
for rep in range(8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    eng.check(lib.ba_problem_triangulate(sh.handle, c, c + K24, float(sc['phi_x']), 0, eng.ptr(Pw), eng.ptr(Xo), eng.stream_ptr()))
    b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
t = float(np.median(ts[2:])); byts = 20 * sc['i'].size + 8 * (N + 1) + 24 * N
print(f'ba_problem_triangulate kernels {t:8.1f} us  {byts / t / 1e6:6.2f} TB/s = {byts / t / 1e6 / 8 * 100:5.1f} % of 8 TB/s')
