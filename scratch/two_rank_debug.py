import faulthandler, importlib, os, sys, time
import numpy as np
import torch, torch.distributed as dist, torch.multiprocessing as mp
ROOT = os.environ.get('GRAFT_REPO_ROOT', '.')
def worker(rank, world, port):
    log = open(os.path.join(ROOT, 'gpurun_out', f'dbg_rank{rank}.log'), 'w', buffering=1)
    faulthandler.dump_traceback_later(25, repeat=False, file=log, exit=True)
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    print('pg up', file=log)
    ba = importlib.import_module('3d_reconstruction_amd')
    g = np.load(os.path.join(ROOT, 'tests/golden/f6_medium.npz'), allow_pickle=False)
    t = torch.ones(4, device='cuda:0', dtype=torch.float64)
    dist.all_reduce(t); print('allreduce ok', t.tolist(), file=log)
    lines = []
    mode = os.environ.get('DBG_MODE', '')
    fl = importlib.import_module('3d_reconstruction_amd.fit_loop')
    orig = fl.SceneFit.__init__
    def patched(self, *a, **k):
        if 'nospec' in mode: k['speculate'] = False
        if 'nofuse' in mode: k['fuse'] = False
        orig(self, *a, **k)
    fl.SceneFit.__init__ = patched
    out = ba.fit_scene_sparse(g['obs_i'], g['obs_j'], g['obs_xy'] + 0.5, int(g['N']), int(g['K']), int(g['W']), int(g['H']),
                              1e-3, 120, g['X_0'], g['C_0'], g['Theta_0'], float(g['phi_x_0']), patience=100, print_step=50, device='cuda:0')
    print('adam ok', out[-1][:3], file=log)
    sc = ba.scenes.orbit_scene(N=300, K=80, track_len=8, seed=4)
    lm = ba.fit_scene_sparse(sc['i'], sc['j'], sc['uv'], sc['N_local'], sc['K'], sc['W'], sc['H'], 1e-3, 6,
                             sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'], optimizer='LM', print_step=1, device='cuda:0')
    print('lm ok', lm[4], file=log)
    dist.destroy_process_group()
if __name__ == '__main__':
    mp.spawn(worker, args=(2, 29917), nprocs=2, join=True)
