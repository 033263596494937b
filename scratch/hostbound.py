"""Host time per iteration: a scene so small that the GPU is never the limit."""
import importlib, sys, os, time, torch
import torch.distributed as dist
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29534')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
ba = importlib.import_module('3d_reconstruction_amd')
sc = ba.scenes.orbit_scene(N=20000, K=200, track_len=10, seed=0)
for forced in (False, True):
    sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], sc['K'], sc['W'], sc['H'], 'cuda:0')
    sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    if forced:
        sh.distributed = True
    fit = ba.SceneFit(sh, lr=1e-3, max_iters=5000, patience=100000, print_step=10 ** 9, printer=lambda *a: None)
    for _ in range(50):
        fit.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        fit.step()
    torch.cuda.synchronize()
    print(f'forced={forced} fuse={fit.fuse}: {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us per iteration (200 k observations)', flush=True)
    sh.close()
dist.destroy_process_group()
