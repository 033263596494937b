"""Set-up time of a Shard (second creation in a process: no module loading): setup_time.py"""
import importlib, os, sys, time, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
ba = importlib.import_module('3d_reconstruction_amd')
sc = ba.scenes.orbit_scene(N=1_000_000, K=2000, track_len=10, seed=0)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], 2000, sc['W'], sc['H'], 'cuda:0')
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f'Shard() {1e3 * (t1 - t0):.1f} ms', getattr(sh, 'setup_ms', None), flush=True)
    sh.close()
