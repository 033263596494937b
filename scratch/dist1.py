"""Host/launch overhead of the multi-GPU sequence, rehearsed on a one-rank RCCL group (the all-reduce moves nothing)."""
import importlib, sys, os, time, numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
ba = importlib.import_module('3d_reconstruction_amd')
sc = ba.scenes.orbit_scene(N=1_000_000, K=2000, track_len=10, seed=0)
for forced, how in ((False, ''), (True, 'node exchange'), (True, 'RCCL on the stream'), (True, 'torch.distributed')):
    os.environ.pop('BA_NODE_EXCHANGE', None); os.environ.pop('BA_TORCH_ALLREDUCE', None)
    if how == 'node exchange': os.environ['BA_NODE_EXCHANGE'] = '1'
    if how == 'torch.distributed': os.environ['BA_TORCH_ALLREDUCE'] = '1'
    sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], sc['K'], sc['W'], sc['H'], 'cuda:0')
    sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    if forced:
        sh.distributed = True
        sh._stage_reduce = False
    fit = ba.SceneFit(sh, lr=1e-3, max_iters=1400, patience=1000, print_step=10 ** 9, printer=lambda *a: None)
    for _ in range(900):
        fit.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        fit.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print(f'distributed sequence forced={forced} {how}: {dt*1e6:.1f} us per iteration, E={fit.E[1099]:.12f} [{sh.rank_sum_path}]', flush=True)
    sh.close()
dist.destroy_process_group()
