"""Time stamps (100 MHz wall clock) inside the camera step of an experimental build: where its 20 us go."""
import importlib, sys, os, ctypes, numpy as np, torch
sys.path.insert(0, '.')
ba = importlib.import_module('3d_reconstruction_amd')
L = importlib.import_module('3d_reconstruction_amd._lib')
L.LIB_PATH = os.path.abspath(os.environ['BA_LIB'])
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
sc = ba.scenes.orbit_scene(N=1_000_000, K=K, track_len=10, seed=0)
sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], sc['K'], sc['W'], sc['H'], 'cuda:0')
sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
fit = ba.SceneFit(sh, lr=1e-3, max_iters=200, patience=1000, print_step=10 ** 9, printer=lambda *a: None)
lib = L.load()
lib.ba_debug_stage.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
rows = []
for it in range(60):
    fit.step()
    torch.cuda.synchronize()
    buf = (ctypes.c_double * 6)()
    lib.ba_debug_stage(sh.handle, buf)
    rows.append([buf[k] for k in range(6)])
r = np.array(rows[20:])
d = (r - r[:, :1]) * 10.0      # ns
print('stamps after kernel entry of workgroup 0 [ns]: entry, loads+gradient, atan2 done, last ticket drawn, factor known, end')
print(np.median(d, axis=0))
