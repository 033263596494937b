"""cfg3 workload for counter collection: queued Adam iterations (fused gradient pass), J^T J passes, one LM step."""
import importlib, sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
ba = importlib.import_module('3d_reconstruction_amd')
sc = ba.scenes.orbit_scene(N=1_000_000, K=2000, track_len=10, seed=0)
sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], sc['K'], sc['W'], sc['H'], 'cuda:0')
sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
fit = ba.SceneFit(sh, lr=1e-3, max_iters=100, patience=1000, print_step=10 ** 9, printer=lambda *a: None)
for _ in range(12):
    fit.step()
torch.cuda.synchronize()
for _ in range(3):
    sh.eval_jtj()
if len(sys.argv) > 1 and sys.argv[1] == 'lm':
    sh.lm_step(1e-3)
torch.cuda.synchronize()
sh.close()
