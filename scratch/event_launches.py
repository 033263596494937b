"""The event-timed launches bench.py takes behind its timed loop, one by one: event_launches.py [repeats]"""
import importlib, importlib.util, os, sys, numpy as np, torch
root = os.environ.get('GRAFT_REPO_ROOT', '.')
sys.path.insert(0, root)
spec = importlib.util.spec_from_file_location('bench', os.path.join(root, 'bench.py')); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
ba = importlib.import_module('3d_reconstruction_amd')
sc = ba.scenes.orbit_scene(N=1_000_000, K=2000, track_len=10, seed=0)
sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], sc['K'], sc['W'], sc['H'], 'cuda:0')
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    fit, dt = bench.run_loop(ba, sh, sc, 300, 300, torch.cuda.synchronize)
    k = np.array(sh.eval_kernel_ms()) * 1e3
    print(f'loop {dt / 300 * 1e6:.1f} us per iteration; event-timed launches (us): ' + ' '.join(f'{v:.1f}' for v in k) + f'  mean {k.mean():.1f} median {np.median(k):.1f}', flush=True)
