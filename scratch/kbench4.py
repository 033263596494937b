"""Observation-pass kernel variants at cfg3: median kernel time from the library's HIP events."""
import importlib, sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
ba = importlib.import_module('3d_reconstruction_amd')
_lib = importlib.import_module('3d_reconstruction_amd._lib')
sc = ba.scenes.orbit_scene(N=1_000_000, K=2000, track_len=10, seed=0)
N, K = sc['N_local'], sc['K']
nv = sc['i'].size
algo = 24 * nv + 48 * N + 96 * K
def bench(tag):
    sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], N, K, sc['W'], sc['H'], 'cuda:0')
    sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    for name, kw in (('camera-runs', {}), ('point-major', dict(point_major=True))):
        for _ in range(23):
            sh.eval(timed=True, **kw)
        torch.cuda.synchronize()
        ms = sh.eval_kernel_ms()[3:]
        m = float(np.median(ms))
        print(f'{tag:24s} {name:12s} median {m*1e3:8.1f} us  min {min(ms)*1e3:7.1f}  {algo/m/1e6:8.1f} GB/s  frac {algo/m/1e6/8000:.3f}  E={sh.energy():.9f}', flush=True)
    sh.close()
for envs in (sys.argv[1:] or ['BA_VERBOSE=1']):
    for kv in envs.split(','):
        k, v = kv.split('=')
        os.environ[k] = v
    bench(envs)
