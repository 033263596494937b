"""Wall time of whole LM iterations (lm.LMFit) at cfg3 / cfg5 sizes: lm_loop.py [K] [N] [iterations]"""
import importlib, os, sys, time, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
if os.environ.get('BA_LIB'):
    importlib.import_module('3d_reconstruction_amd._lib').LIB_PATH = os.path.abspath(os.environ['BA_LIB'])
import torch
ba = importlib.import_module('3d_reconstruction_amd')
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 8
sc = ba.scenes.orbit_scene(N=N, K=K, track_len=10, seed=0)
sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], K, sc['W'], sc['H'], 'cuda:0')
sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
lines = []
fit = ba.lm.LMFit(sh, iters + 1, lam=1e-3, print_step=10 ** 9, printer=lines.append)
torch.cuda.synchronize(); t0 = time.perf_counter()
fit.run()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
E = fit.energies()
st = np.diff(np.array(fit.stamps)) * 1e3      # [first evaluation, iteration 1 (allocates and zeroes S, ...), iteration 2, ...]
later = st[2:] if len(st) > 2 else st[1:]
print(f'K={K} N={N} |V|={sc["i"].size}: {fit.iters} LM iterations ({fit.accepted} accepted, {fit.rejected} rejected) in {dt*1e3:.1f} ms; '
      f'first evaluation {st[0]:.1f} ms, first iteration {st[1]:.1f} ms (with the set-up of S), the others {later.mean():.2f} ms each '
      f'(min {later.min():.2f}, max {later.max():.2f}); E ' + ' '.join(f'{e:.5g}' for e in E), flush=True)
import ctypes as C
buf, nl = (C.c_int64 * 16)(), C.c_int(0)
sh.lib.ba_cholesky_last_plan(buf, 4, C.byref(nl))
print('plan (first block, chains, blocks per chain, separator blocks):', [tuple(buf[4 * k:4 * k + 4]) for k in range(nl.value)], 'of', sh.ld_red // 64, 'blocks', flush=True)
