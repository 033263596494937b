import importlib, sys, os, numpy as np, torch, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
if os.environ.get('BA_LIB'):
    importlib.import_module('3d_reconstruction_amd._lib').LIB_PATH = os.path.abspath(os.environ['BA_LIB'])
ba = importlib.import_module('3d_reconstruction_amd')
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
sc = ba.scenes.orbit_scene(N=N, K=K, track_len=10, seed=0)
sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], sc['K'], sc['W'], sc['H'], 'cuda:0')
sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
sh.gauge_fix()
nv = sc['i'].size
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
for _ in range(6): sh.eval_jtj(timed=True)
torch.cuda.synchronize()
ms = float(np.median(sh.eval_kernel_ms()[1:]))
algo = 24*nv + 24*N + 24*N + 48*N + 24*N + 96*K + 216*K
print(f'k_jtj_tiled: {ms*1e3:.1f} us, algorithmic {algo/1e6:.1f} MB -> {algo/ms/1e6:.1f} GB/s = {algo/ms/1e6/8000:.3f} of 8 TB/s', flush=True)
import ctypes as C
_lib = ba.engine._lib
p = _lib.ptr
lam = 1e-3
def schur():
    _lib.check(sh.lib.ba_schur_build(sh.handle, p(sh.X), lam, p(sh.V), p(sh.Xphi), p(sh.gX), p(sh.B), p(sh.bphi), p(sh.hphi), p(sh.gC), p(sh.gTheta), p(sh.gphi), 1, p(sh.S), p(sh.rhs), _lib.stream_ptr()))
t_s = timeit(schur, 3)
print(f'schur build: {t_s:.2f} ms', flush=True)
n, ld = sh.n_red, sh.ld_red
S0, r0 = sh.S.clone(), sh.rhs.clone()
info = C.c_int(0)
def chol():
    sh.S.copy_(S0); sh.rhs.copy_(r0)
    rc = sh.lib.ba_cholesky_solve(n, ld, p(sh.S), p(sh.rhs), C.byref(info), _lib.stream_ptr())
    if not os.environ.get('BA_ABL'): _lib.check(rc)
def copy_only():
    sh.S.copy_(S0); sh.rhs.copy_(r0)
t_c = timeit(chol, 3) - timeit(copy_only, 3)
flops = n**3/3
print(f'cholesky+solve n={n}: {t_c:.2f} ms -> {flops/t_c/1e9:.2f} TFLOP/s fp64 ({flops/t_c/1e9/78.6:.3f} of 78.6 TF matrix peak), info={info.value}', flush=True)
# check solution residual
chol(); torch.cuda.synchronize()
x = sh.rhs[:n].clone()
Sl = torch.tril(S0[:n,:n]); Sfull = Sl + torch.tril(S0[:n,:n], -1).T
res = (Sfull @ x - r0[:n]).abs().max().item() / r0[:n].abs().max().item()
print('relative residual of the solve', res)
t0=time.time(); out = None
