"""J^T J pass and gradient pass (stand-alone) at several shard sizes: fraction of the HBM roofline by the algorithmic bytes."""
import importlib, sys, numpy as np, torch
sys.path.insert(0, '.')
ba = importlib.import_module('3d_reconstruction_amd')
for N in [int(a) for a in sys.argv[1:]] or [1_000_000, 1_250_000, 2_500_000, 10_000_000]:
    K = 2000
    sc = ba.scenes.orbit_scene(N=N, K=K, track_len=10, seed=0)
    nv = sc['i'].size
    sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], N, K, sc['W'], sc['H'], 'cuda:0')
    sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    algo_g = 24 * nv + 48 * N + 96 * K
    algo_j = 24 * nv + 96 * N + 96 * K + 216 * K
    for _ in range(13):
        sh.eval(timed=True)
    torch.cuda.synchronize()
    g = float(np.median(sh.eval_kernel_ms()[3:]))
    for _ in range(13):
        sh.eval_jtj(timed=True)
    torch.cuda.synchronize()
    j = float(np.median(sh.eval_kernel_ms()[3:]))
    print(f'N={N} |V|={nv}: gradient pass {g * 1e3:.1f} us = {algo_g / g / 1e6 / 8000:.3f} of 8 TB/s; '
          f'J^T J pass {j * 1e3:.1f} us = {algo_j / j / 1e6 / 8000:.3f}', flush=True)
    sh.close(); del sh, sc
    torch.cuda.empty_cache()
