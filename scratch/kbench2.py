import importlib, sys, numpy as np, torch, time
sys.path.insert(0, '.')
ba = importlib.import_module('3d_reconstruction_amd')
from importlib import import_module
_lib = import_module('3d_reconstruction_amd._lib')
sc = ba.scenes.orbit_scene(N=1_000_000, K=2000, track_len=10, seed=0)
# sort points by first camera
N = sc['N_local']
first = np.full(N, 1<<30); np.minimum.at(first, sc['i'], sc['j'])
perm = np.argsort(first, kind='stable')         # new order -> old index
inv = np.empty(N, dtype=np.int64); inv[perm] = np.arange(N)
new_i = inv[sc['i']]
order = np.argsort(new_i, kind='stable')
i2 = new_i[order].astype(np.int32); j2 = sc['j'][order]; uv2 = sc['uv'][order]
X2 = sc['X_0'][perm]
def bench(i, j, uv, X, tag):
    sh = ba.engine.Shard(i, j, uv, N, sc['K'], sc['W'], sc['H'], 'cuda:0')
    sh.set_params(X, sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    nv = i.size
    algo = 24*nv + 48*N + 96*sc['K']
    p = _lib.ptr
    for extra, name in [(0,'tiled'), (64,'tiled skip cam'), (128,'tiled skip pt'), (192,'tiled skip both'), (512,'untiled')]:
        for _ in range(23):
            _lib.check(sh.lib.ba_eval(sh.handle, p(sh.X), p(sh.C), p(sh.Theta), p(sh.phi), 1|2|4|32|extra, p(sh.E_sum), p(sh.gX), p(sh.gC), p(sh.gTheta), p(sh.gphi), _lib.stream_ptr()))
        torch.cuda.synchronize()
        ms = sh.eval_kernel_ms()[3:]
        m = float(np.median(ms))
        print(f'{tag:10s} {name:12s} median {m*1e3:8.1f} us   {algo/m/1e6:8.1f} GB/s  frac {algo/m/1e6/8000:.3f}  E={sh.energy():.6f}', flush=True)
    sh.close()
bench(sc['i'], sc['j'], sc['uv'], sc['X_0'], 'random')
bench(i2, j2, uv2, X2, 'sorted')
