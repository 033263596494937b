"""Both passes on scenes whose tracks start in groups of `ppc` per camera (regular: exactly ppc, in point order; random: random
centre frames, about ppc per camera) at a given tile size: regular_tiles.py ppc:tile_obs:K[:r|:tN] ...   (r: random centres, tN: track length N instead of 10)"""
import importlib, os, sys, numpy as np, torch
os.environ['BA_VERBOSE'] = '1'
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
ba = importlib.import_module('3d_reconstruction_amd')
for spec in sys.argv[1:]:
    f = spec.split(':')
    ppc, tobs, K, rnd = int(f[0]), int(f[1]), int(f[2]), 'r' in f[3:]
    track = ([int(x[1:]) for x in f[3:] if x.startswith('t')] or [10])[0]
    N = ppc * K
    os.environ['BA_TILE_OBS'] = str(tobs)
    sc = ba.scenes.orbit_scene(N=N, K=K, track_len=track, seed=0, centres=None if rnd else np.arange(N) // ppc)
    nv = sc['i'].size
    sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], N, K, sc['W'], sc['H'], 'cuda:0')
    sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    algo_g = 24 * nv + 48 * N + 96 * K
    algo_j = 24 * nv + 96 * N + 96 * K + 216 * K
    for _ in range(23):
        sh.eval(timed=True)
    torch.cuda.synchronize()
    g = float(np.median(sh.eval_kernel_ms()[3:]))
    for _ in range(23):
        sh.eval_jtj(timed=True)
    torch.cuda.synchronize()
    j = float(np.median(sh.eval_kernel_ms()[3:]))
    print(f'{spec}: N={N} |V|={nv}: gradient pass {g * 1e3:.1f} us = {algo_g / g / 1e6 / 8000:.3f}; '
          f'J^T J pass {j * 1e3:.1f} us = {algo_j / j / 1e6 / 8000:.3f}; ps per obs: {g * 1e9 / nv:.2f} / {j * 1e9 / nv:.2f}', flush=True)
    sh.close(); del sh, sc
    torch.cuda.empty_cache()
