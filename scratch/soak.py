"""Long run of the fused loop at cfg3 with a learning rate that makes the schedule roll back now and then:
two identical runs, their event counts and how far their energies drift apart (atomics: not bitwise)."""
import importlib, sys, os, time, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
ba = importlib.import_module('3d_reconstruction_amd')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 3e-2
sc = ba.scenes.orbit_scene(N=1_000_000, K=2000, track_len=10, seed=0)
sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], sc['K'], sc['W'], sc['H'], 'cuda:0')
runs = []
for rep in range(2):
    sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    log = []
    fit = ba.SceneFit(sh, lr=lr, max_iters=iters, patience=300, print_step=10 ** 9, printer=lambda *a: log.append(' '.join(str(x) for x in a)))
    t0 = time.perf_counter(); n = 0
    while fit.step():
        n += 1
        if n % 10000 == 0:
            print(f'run {rep}: {n} steps, {time.perf_counter() - t0:.1f} s, E={fit.E[fit.iters - 1]:.9e} lr={fit.lr:.3e} dropped={fit.dropped}', flush=True)
    torch.cuda.synchronize()
    E = np.array(fit.E[:fit.iters], dtype=np.float64)
    print(f'run {rep}: iters={fit.iters} time={time.perf_counter() - t0:.2f} s  E0={E[0]:.6e} Emin={E.min():.9e} Elast={E[-1]:.9e} finite={np.isfinite(E).all()} '
          f'lr_end={fit.lr:.3e} dropped={fit.dropped} messages={len(log)}', flush=True)
    for m in log[:6]:
        print('   ', m[:150])
    runs.append(E)
n = min(len(runs[0]), len(runs[1]))
rel = np.abs(runs[0][:n] - runs[1][:n]) / np.abs(runs[0][:n])
for k in (10, 100, 1000, 10000, n - 1):
    if k < n:
        print(f'relative energy difference between the runs up to iteration {k}: {rel[:k + 1].max():.3e}')
