"""Where the host's time goes in the fused loop on a small scene (cProfile of 3000 iterations) and the GPU's share
(kernel trace beside it: rocprofv3 --kernel-trace --stats -- python3 scratch/small_profile.py)."""
import cProfile, importlib, os, pstats, sys, time, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
ba = importlib.import_module('3d_reconstruction_amd')
sc = ba.scenes.orbit_scene(N=100, K=300, track_len=300, seed=0)
sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], sc['K'], sc['W'], sc['H'], 'cuda:0')
sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
fit = ba.SceneFit(sh, lr=1e-3, max_iters=5100, patience=10 ** 7, print_step=10 ** 9, printer=lambda *a: None)
for _ in range(1000): fit.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(1000): fit.step()
torch.cuda.synchronize(); print('%.1f us per iteration' % ((time.perf_counter() - t0) / 1000 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(3000): fit.step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
