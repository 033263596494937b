"""Cost of a timed launch: iteration time of the fused loop with every 8th / no launch timed (BA_LIB selects the library)."""
import importlib, sys, os, time, numpy as np, torch
sys.path.insert(0, '.')
ba = importlib.import_module('3d_reconstruction_amd')
if os.environ.get('BA_LIB'):
    importlib.import_module('3d_reconstruction_amd._lib').LIB_PATH = os.path.abspath(os.environ['BA_LIB'])
sc = ba.scenes.orbit_scene(N=1_000_000, K=2000, track_len=10, seed=0)
sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], sc['K'], sc['W'], sc['H'], 'cuda:0')
for every in (0, 8, 0, 8):
    sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    fit = ba.SceneFit(sh, lr=1e-3, max_iters=500, patience=100000, print_step=10 ** 9, printer=lambda *a: None)
    fit.timed_every = every
    for _ in range(40):
        fit.step()
    sh.eval_kernel_ms()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(240):
        fit.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 240 * 1e6
    ms = sh.eval_kernel_ms()
    print(os.environ.get('BA_LIB', 'built'), f'timed_every={every}: {dt:.1f} us per iteration' + (f', kernel {np.median(ms) * 1e3:.1f} us ({len(ms)} samples)' if ms else ''), flush=True)
