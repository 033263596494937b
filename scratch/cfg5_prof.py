import importlib, sys, time, torch
sys.path.insert(0, '.')
ba = importlib.import_module('3d_reconstruction_amd')
sc = ba.scenes.orbit_scene(N=1_000_000, K=20000, track_len=10, seed=0)
sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], sc['K'], sc['W'], sc['H'], 'cuda:0')
sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
fit = ba.SceneFit(sh, lr=1e-3, max_iters=400, patience=100000, print_step=10 ** 9, printer=lambda *a: None)
for _ in range(130):
    fit.step()
torch.cuda.synchronize()
