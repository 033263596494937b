"""Iteration time of the fused loop (cfg3, untimed) with the library given by BA_LIB (default: the built one)."""
import importlib, sys, os, time, numpy as np, torch
sys.path.insert(0, '.')
ba = importlib.import_module('3d_reconstruction_amd')
if os.environ.get('BA_LIB'):
    importlib.import_module('3d_reconstruction_amd._lib').LIB_PATH = os.path.abspath(os.environ['BA_LIB'])
sc = ba.scenes.orbit_scene(N=1_000_000, K=2000, track_len=10, seed=0)
sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], sc['K'], sc['W'], sc['H'], 'cuda:0')
out = []
for rep in range(3):
    sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    fit = ba.SceneFit(sh, lr=1e-3, max_iters=500, patience=100000, print_step=10 ** 9, printer=lambda *a: None)
    for _ in range(30):
        fit.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300):
        fit.step()
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / 300 * 1e6)
print(os.environ.get('BA_LIB', 'built'), ' '.join(f'{x:.1f}' for x in out), f'E={fit.E[fit.iters - 1]:.12f}')
