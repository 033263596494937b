"""Fused iteration time and the pass kernel's time for BA_LIB: ab_iter.py [points] [cameras]"""
import importlib, os, sys, time, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
if os.environ.get('BA_LIB'):
    importlib.import_module('3d_reconstruction_amd._lib').LIB_PATH = os.path.abspath(os.environ['BA_LIB'])
import torch
ba = importlib.import_module('3d_reconstruction_amd')
tag = os.path.basename(os.environ.get('BA_LIB', 'built'))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
sc = ba.scenes.orbit_scene(N=N, K=K, track_len=10, seed=0)
sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], N, K, sc['W'], sc['H'], 'cuda:0')
sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
fit = ba.SceneFit(sh, lr=1e-3, max_iters=5000, patience=100000, print_step=10 ** 9, printer=lambda *a: None)
fit.timed_every = 8
for _ in range(500):
    fit.step()
torch.cuda.synchronize(); sh.eval_kernel_ms()
out = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(400):
        fit.step()
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / 400 * 1e6)
k = float(np.mean(sh.eval_kernel_ms())) * 1e3
print(f'{tag:22s} iteration ' + ' '.join(f'{x:.1f}' for x in out) + f' us; pass kernel {k:.1f} us; E={fit.E[fit.iters - 1]:.6f}', flush=True)
