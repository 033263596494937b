"""Host time of the calls of one LM step (queue empty before each): lm_host_time.py"""
import importlib, os, sys, time, numpy as np, ctypes as C
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
ba = importlib.import_module('3d_reconstruction_amd')
sc = ba.scenes.orbit_scene(N=1_000_000, K=2000, track_len=10, seed=0)
sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], 2000, sc['W'], sc['H'], 'cuda:0')
sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
sh.gauge_fix(); sh.eval_jtj(); sh.energy()
p = ba.engine.ptr
lam = 1e-3
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sh.lib.ba_schur_build_tracked(sh.handle, sh.X.data_ptr(), lam, p(sh.V), p(sh.Xphi), p(sh.gX), p(sh.B), p(sh.bphi), p(sh.hphi),
                                  p(sh.gC), p(sh.gTheta), p(sh.gphi), 1, p(sh.S), p(sh.rhs), p(sh.S_first), 1 if rep else 0, sh.stream)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    sh.lib.ba_cholesky_solve_queued(sh.n_red, sh.ld_red, p(sh.S), p(sh.rhs), p(sh.S_first), 1, sh.stream)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    sh.eval_jtj()
    t5 = time.perf_counter()
    E = sh.energy()
    t6 = time.perf_counter()
    print(f'schur build call {1e6*(t1-t0):.0f} us (device {1e6*(t2-t0):.0f}); cholesky queued call {1e6*(t3-t2):.0f} us (device {1e6*(t4-t2):.0f}); '
          f'eval_jtj call {1e6*(t5-t4):.0f} us, energy read {1e6*(t6-t5):.0f} us', flush=True)
