"""Graded tail of the tile list (BA_TILE_TAIL="half,quarter": fractions of the observations cut into tiles of a half / a quarter
of the size): both passes stand-alone and the fused iteration.  tail_sweep.py N head/tail [head/tail ...]   e.g. 256:2560,256:1280/0.1"""
import importlib, os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
ba = importlib.import_module('3d_reconstruction_amd')
N, K = int(sys.argv[1]), 2000
sc = ba.scenes.orbit_scene(N=N, K=K, track_len=10, seed=0)
nv = sc['i'].size
algo_g = 24 * nv + 48 * N + 96 * K
algo_j = 24 * nv + 96 * N + 96 * K + 216 * K
for tail in sys.argv[2:]:
    head, _, tl = tail.partition('/')
    os.environ['BA_TILE_HEAD'], os.environ['BA_TILE_TAIL'] = head, tl or '0'
    sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], N, K, sc['W'], sc['H'], 'cuda:0')
    sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    for _ in range(33):
        sh.eval(timed=True)
    torch.cuda.synchronize()
    g = float(np.median(sh.eval_kernel_ms()[3:]))
    for _ in range(33):
        sh.eval_jtj(timed=True)
    torch.cuda.synchronize()
    j = float(np.median(sh.eval_kernel_ms()[3:]))
    its = []
    for rep in range(2):
        sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
        fit = ba.SceneFit(sh, lr=1e-3, max_iters=700, patience=100000, print_step=10 ** 9, printer=lambda *a: None)
        for _ in range(200):
            fit.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(400):
            fit.step()
        torch.cuda.synchronize()
        its.append((time.perf_counter() - t0) / 400 * 1e6)
    print(f'N={N} tail={tail}: gradient pass {g * 1e3:.1f} us = {algo_g / g / 1e6 / 8000:.3f}; J^T J pass {j * 1e3:.1f} us = '
          f'{algo_j / j / 1e6 / 8000:.3f}; fused iteration {its[0]:.1f} {its[1]:.1f} us; E={fit.E[fit.iters - 1]:.12f}', flush=True)
    sh.close(); del sh, fit
    torch.cuda.empty_cache()
