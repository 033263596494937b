"""Timing of the camera-pair homography kernel at cfg3 size (1 M points, 2 k cameras, 10 M observations)."""
import importlib, sys, time
import numpy as np, torch
import os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
ba = importlib.import_module('3d_reconstruction_amd')
sc = ba.scenes.orbit_scene(1_000_000, 2000, track_len=10, seed=0)
print('obs', sc['i'].size, flush=True)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = ba.engine.homography_diff_matrix(sc['i'], sc['j'], sc['uv'], sc['N'], sc['K'])
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f'whole call {1e3 * (t1 - t0):.1f} ms, pairs with a value {int((~torch.isnan(out)).sum()) // 2}', flush=True)
