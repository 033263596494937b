"""A/B of host-side options of the fused loop inside one process (boxes differ by a few per cent): us per iteration."""
import importlib, sys, os, time, numpy as np, torch
sys.path.insert(0, '.')
ba = importlib.import_module('3d_reconstruction_amd')
sc = ba.scenes.orbit_scene(N=1_000_000, K=2000, track_len=10, seed=0)
sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], sc['K'], sc['W'], sc['H'], 'cuda:0')
def run(prepare, n=300):
    sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    fit = ba.SceneFit(sh, lr=1e-3, max_iters=n + 100, patience=100000, print_step=10 ** 9, printer=lambda *a: None)
    if not prepare:
        fit._prepare = None
    for _ in range(30):
        fit.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fit.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
res = {True: [], False: []}
for rep in range(4):
    for prepare in (True, False):
        res[prepare].append(run(prepare))
for k, v in res.items():
    print(f'prepare_buffers={k}: ' + ' '.join(f'{x:.1f}' for x in v) + f'  median {np.median(v):.1f} us')
