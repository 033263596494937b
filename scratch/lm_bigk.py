"""cfg5: large-K scene (K = 20 k cameras, n = 120 001 reduced unknowns, S = 115 GB): one LM step."""
import importlib, sys, os, time, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
ba = importlib.import_module('3d_reconstruction_amd')
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
t0 = time.time()
sc = ba.scenes.orbit_scene(N=N, K=K, track_len=10, seed=0)
print(f'scene N={sc["N_local"]} K={K} nV={sc["i"].size} built in {time.time()-t0:.1f} s', flush=True)
sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], K, sc['W'], sc['H'], 'cuda:0')
sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
sh.gauge_fix()
sh.eval_jtj(timed=True); torch.cuda.synchronize()
E0 = sh.energy()
print(f'E0 = {E0:.6f}; J^T J pass {sh.eval_kernel_ms()[-1]*1e3:.1f} us; reduced system n = {sh.n_red}, S = {sh.ld_red**2*8/1e9:.1f} GB', flush=True)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
info = sh.lm_step(1e-3)
ev[1].record(); torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1])
n = sh.n_red
print(f'lm_step (Schur build + Cholesky + solves + back-substitution): {ms/1e3:.2f} s, info={info}; n^3/3 = {n**3/3/1e12:.1f} TFLOP -> {n**3/3/ms/1e9:.1f} TFLOP/s overall', flush=True)
sh.X.add_(sh.dX); sh.cam.add_(sh.dcam); sh.gauge_fix(); sh.eval()
print(f'E after the step = {sh.energy():.6f} (was {E0:.6f})', flush=True)
