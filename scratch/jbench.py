"""J^T J pass variants at cfg3: median kernel time from the library's HIP events."""
import importlib, sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
ba = importlib.import_module('3d_reconstruction_amd')
sc = ba.scenes.orbit_scene(N=1_000_000, K=2000, track_len=10, seed=0)
N, K = sc['N_local'], sc['K']
nv = sc['i'].size
algo = 24 * nv + 24 * N + 24 * N + 48 * N + 24 * N + 96 * K + 216 * K
def bench(tag, both):
    sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], N, K, sc['W'], sc['H'], 'cuda:0')
    sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    for name, kw in (('camera-runs', {}), ('point-major', dict(point_major=True)))[:2 if both else 1]:
        for _ in range(9):
            sh.eval_jtj(timed=True, **kw)
        torch.cuda.synchronize()
        ms = sh.eval_kernel_ms()[2:]
        m = float(np.median(ms))
        print(f'{tag:28s} {name:12s} median {m*1e3:8.1f} us  min {min(ms)*1e3:7.1f}  {algo/m/1e6:8.1f} GB/s  frac {algo/m/1e6/8000:.3f}  E={sh.energy():.9f} B={float(sh.B.sum()):.9e} V={float(sh.V.sum()):.9e}', flush=True)
    sh.close()
for n, envs in enumerate(sys.argv[1:] or ['BA_VERBOSE=1']):
    for kv in envs.split(','):
        k, v = kv.split('=')
        os.environ[k] = v
    bench(envs, n == 0)
