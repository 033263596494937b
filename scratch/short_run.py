"""Why does a 5 + 20 iteration measurement after the pre-heat differ from the steady state?"""
import importlib, os, sys, time, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
ba = importlib.import_module('3d_reconstruction_amd')
sc = ba.scenes.orbit_scene(N=1_000_000, K=2000, track_len=10, seed=0)
sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], sc['K'], sc['W'], sc['H'], 'cuda:0')
def fresh(max_iters):
    sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    return ba.SceneFit(sh, lr=1e-3, max_iters=max_iters, patience=10 ** 7, print_step=10 ** 9, printer=lambda *a: None)
f0 = fresh(800)
for _ in range(700): f0.step()
torch.cuda.synchronize()
for timed_every, warm, steps in ((0, 5, 20), (10, 5, 20), (20, 5, 20), (0, 5, 20), (0, 50, 20), (0, 5, 100), (0, 300, 300)):
    fit = fresh(warm + steps + 8)
    fit.timed_every = timed_every
    for _ in range(warm): fit.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    per = []
    for _ in range(steps):
        a = time.perf_counter(); fit.step(); per.append((time.perf_counter() - a) * 1e6)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps * 1e6
    print(f'timed_every={timed_every} warm={warm} steps={steps}: {dt:.1f} us/iteration; host step() first ten: ' + ' '.join(f'{x:.0f}' for x in per[:10]) + f'; dropped {fit.dropped}', flush=True)
    sh.eval_kernel_ms()
