import importlib, sys, numpy as np, torch, time
sys.path.insert(0, '.')
ba = importlib.import_module('3d_reconstruction_amd')
from importlib import import_module
_lib = import_module('3d_reconstruction_amd._lib')
sc = ba.scenes.orbit_scene(N=1_000_000, K=2000, track_len=10, seed=0)
sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], sc['K'], sc['W'], sc['H'], 'cuda:0')
sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
nv = sc['i'].size
algo = 24*nv + 48*sc['N_local'] + 96*sc['K']
def run(extra, name, reps=20):
    from ctypes import c_void_p
    lib = sh.lib
    p = _lib.ptr
    for _ in range(reps+3):
        flags = 1|2|4|32|extra
        _lib.check(lib.ba_eval(sh.handle, p(sh.X), p(sh.C), p(sh.Theta), p(sh.phi), flags, p(sh.E_sum), p(sh.gX), p(sh.gC), p(sh.gTheta), p(sh.gphi), _lib.stream_ptr()))
    torch.cuda.synchronize()
    ms = sh.eval_kernel_ms()[3:]
    m = float(np.median(ms))
    print(f'{name:40s} median {m*1e3:8.1f} us  min {min(ms)*1e3:8.1f} us   {algo/m/1e6:8.1f} GB/s  frac {algo/m/1e6/8000:.3f}', flush=True)
variants = [(0,'default (LDS cam)'), (16,'global atomics cam'), (64,'skip cam accumulation'), (128,'skip point scan'), (192,'skip both')]
if len(sys.argv) > 1:
    variants = [(int(a), 'flags+%s' % a) for a in sys.argv[1:]]
for extra, name in variants:
    run(extra, name)
