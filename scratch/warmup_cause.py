"""Why are the first iterations of a fit slower?  Kernel time of the pass (events handed to the dispatch, every
launch) in windows of 50 iterations for: a fresh fit; the same fit continued after 2 s of idling; a second fresh fit
on the same shard from the same start; a fit started from the parameters the first one reached."""
import importlib, os, sys, time, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
ba = importlib.import_module('3d_reconstruction_amd')
sc = ba.scenes.orbit_scene(N=1_000_000, K=2000, track_len=10, seed=0)
sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], sc['K'], sc['W'], sc['H'], 'cuda:0')

def windows(fit, n, label):
    out = []
    for w in range(n // 50):
        for _ in range(50):
            fit.step()
        torch.cuda.synchronize()
        out.append(float(np.mean(sh.eval_kernel_ms())) * 1e3)
    print(f'{label:46s}', ' '.join(f'{x:5.1f}' for x in out), f' E={fit.E[fit.iters - 1]:.4f}', flush=True)

def new_fit(X, C, Th, phi):
    sh.set_params(X, C, Th, phi)
    fit = ba.SceneFit(sh, lr=1e-3, max_iters=5000, patience=100000, print_step=10 ** 9, printer=lambda *a: None)
    fit.timed_every = 1
    return fit

fit = new_fit(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
windows(fit, 500, 'fresh fit, iterations 0..499')
time.sleep(2.0)
windows(fit, 200, 'same fit after 2 s idle, iterations 500..699')
X1 = sh.points_to_caller(sh.X).cpu().numpy().copy()
C1, T1, p1 = sh.C.cpu().numpy().copy(), sh.Theta.cpu().numpy().copy(), float(sh.phi.item())
fit2 = new_fit(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
windows(fit2, 400, 'second fresh fit from the same start')
fit3 = new_fit(X1, C1, T1, p1)
windows(fit3, 300, 'fresh fit (zero moments) from the reached state')
