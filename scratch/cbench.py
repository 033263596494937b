"""Camera step kernel timing (K cameras) via torch events; BA_DBG_CAM selects truncated variants."""
import importlib, sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
ba = importlib.import_module('3d_reconstruction_amd')
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
sc = ba.scenes.orbit_scene(N=20000, K=K, track_len=10, seed=0)
for dbg in ('0', '1', '2'):
    os.environ['BA_DBG_CAM'] = dbg
    sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], K, sc['W'], sc['H'], 'cuda:0')
    sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    sh.gauge_fix()
    sh.eval()
    mc = (torch.zeros_like(sh.cam), torch.zeros_like(sh.cam))
    ts = []
    for it in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            sh.camera_update(2, 1e-3, 0.1, 0.001, (0, 6 * K + 1), mc, True)
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / 10)
    print(f'K={K} BA_DBG_CAM={dbg}: camera step {np.median(ts)*1e3:.1f} us per launch (10 back-to-back)', flush=True)
    sh.close()
