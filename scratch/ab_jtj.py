"""J^T J pass kernel time only (BA_LIB selects the library): ab_jtj.py [points]"""
import importlib, os, sys, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
if os.environ.get('BA_LIB'):
    importlib.import_module('3d_reconstruction_amd._lib').LIB_PATH = os.path.abspath(os.environ['BA_LIB'])
import torch
ba = importlib.import_module('3d_reconstruction_amd')
tag = os.path.basename(os.environ.get('BA_LIB', 'built'))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
sc = ba.scenes.orbit_scene(N=N, K=2000, track_len=10, seed=0)
sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], N, 2000, sc['W'], sc['H'], 'cuda:0')
sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
out = []
for rep in range(3):
    for _ in range(23):
        sh.eval_jtj(timed=True)
    torch.cuda.synchronize()
    out.append(float(np.median(sh.eval_kernel_ms()[3:])) * 1e3)
g = []
for rep in range(2):
    for _ in range(23):
        sh.eval(timed=True)
    torch.cuda.synchronize()
    g.append(float(np.median(sh.eval_kernel_ms()[3:])) * 1e3)
print(f'{tag:24s} jtj ' + ' '.join(f'{x:.1f}' for x in out) + '   grad ' + ' '.join(f'{x:.1f}' for x in g), flush=True)
