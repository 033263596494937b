"""Two processes on one GPU: many node-exchange calls of changing length queued without host synchronisation, every
sum compared with the exact one on the device."""
import importlib, os, socket, sys, time
import torch, torch.distributed as dist, torch.multiprocessing as mp
ROOT = os.environ.get('GRAFT_REPO_ROOT', '.')

def worker(rank, world, port, calls):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sharding = importlib.import_module('3d_reconstruction_amd.sharding')
    dev = torch.device('cuda:0')
    ex = sharding.NodeExchange.create(None, dev, 20000)
    assert ex is not None
    st = torch.cuda.current_stream().cuda_stream
    lengths = [17210, 12001, 7, 20000, 1024, 1, 5000, 17210, 17210, 3]
    base = torch.arange(1, 20001, dtype=torch.float64, device=dev)
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    t0 = time.perf_counter()
    for c in range(calls):
        n = lengths[(c * 7 + c // 13) % len(lengths)]
        parts = [base[:n] * (1.0 + 0.37 * r) + float(c % 977) * (r + 1) for r in range(world)]
        want = parts[0].clone()
        for r in range(1, world):
            want += parts[r]
        t = parts[rank].clone()
        ex(t, st)
        bad += (t != want).sum()
        if (c + 1) % 5000 == 0:
            torch.cuda.synchronize(); ex.check()
            print(f'rank {rank}: {c + 1} calls, mismatches {int(bad.item())}, {time.perf_counter() - t0:.1f} s', flush=True)
    torch.cuda.synchronize(); ex.check()
    assert int(bad.item()) == 0
    dist.barrier(); ex.close(); dist.destroy_process_group()

if __name__ == '__main__':
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port, int(sys.argv[1]) if len(sys.argv) > 1 else 20000), nprocs=2, join=True)
    print('soak ok')
