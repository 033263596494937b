"""The N > 1 path on CPU: two gloo ranks, each holding a point range of one scene, run the same
host loop (fit_loop.SceneFit + sharding) with the oracle-backed stand-in shard; the camera-side
sums and the energy cross ranks through sharding.allreduce_sum_, exactly as on GPUs through RCCL.
The result must equal the single-process run of the reference (golden F3, LR events included)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from mock_shard import MockShard
        from oracle import ba_oracle as orc
        fit_loop = importlib.import_module('3d_reconstruction_amd.fit_loop')
        sharding = importlib.import_module('3d_reconstruction_amd.sharding')
        g = np.load(os.path.join(GOLDEN, 'f3_lr_events.npz'), allow_pickle=False)
        i, j, uv = orc.visibility_coo(g['X_pix'])
        N, K = g['X_pix'].shape[:2]
        pb, ob = sharding.partition_points(i, N, world)
        p0, p1, e0, e1 = pb[rank], pb[rank + 1], ob[rank], ob[rank + 1]
        sh = MockShard(i[e0:e1] - p0, j[e0:e1], uv[e0:e1], p1 - p0, K, int(g['W']), int(g['H']),
                       n_visible_total=i.size)
        sh.set_params(g['X_0'][p0:p1], g['C_0'], g['Theta_0'], float(g['phi_x_0']))
        lines = []
        fit = fit_loop.SceneFit(sh, 1e-3, 200, patience=20, printer=lines.append)
        fit.run()
        X_full = sharding.all_gather_rows(sh.X, pb)
        # per-iteration history of a sharded run: every snapshot of X is gathered over the ranks
        p2w = importlib.import_module('3d_reconstruction_amd.pix2world')
        g1 = np.load(os.path.join(GOLDEN, 'f1_tiny_masks.npz'), allow_pickle=False)
        i1, j1, uv1 = orc.visibility_coo(g1['X_pix'])
        N1, K1 = g1['X_pix'].shape[:2]
        pb1, ob1 = sharding.checked_partition(i1, N1, world)
        q0, q1, f0, f1 = pb1[rank], pb1[rank + 1], ob1[rank], ob1[rank + 1]
        sh1 = MockShard(i1[f0:f1] - q0, j1[f0:f1], uv1[f0:f1], q1 - q0, K1, int(g1['W']), int(g1['H']),
                        n_visible_total=i1.size)
        sh1.set_params(g1['X_0'][q0:q1], g1['C_0'], g1['Theta_0'], float(g1['phi_x_0']))
        fit1 = fit_loop.SceneFit(sh1, 1e-3, 5, ret_all_iters=True, printer=lambda *a: None)
        fit1.run()
        packed = p2w._pack_results(fit1, sh1, (True, True, True), lambda x: sharding.all_gather_rows(x, pb1))
        # a split that would leave a rank without observations is refused on every rank alike
        try:
            sharding.checked_partition(np.zeros(5, dtype=np.int64), 1, world)
            refused = False
        except ValueError:
            refused = True
        np.savez(os.path.join(out_dir, f'rank{rank}.npz'), E=fit.energies(), X=X_full.numpy(), C=sh.C.numpy(),
                 Theta=sh.Theta.numpy(), phi=sh.phi.numpy(), log=np.array(lines), bounds=np.array(pb),
                 hist_X=packed[5], hist_C=packed[6], hist_phi=packed[8], final_X1=packed[0], refused=refused)
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_run_matches_reference(tmp_path):
    from conftest import logs_match, rel_err
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = np.load(os.path.join(GOLDEN, 'f3_lr_events.npz'), allow_pickle=False)
    r0 = np.load(tmp_path / 'rank0.npz')
    r1 = np.load(tmp_path / 'rank1.npz')
    # both ranks took identical decisions and hold identical replicated state
    assert np.array_equal(r0['E'], r1['E']) and np.array_equal(r0['log'], r1['log'])
    assert np.array_equal(r0['C'], r1['C']) and np.array_equal(r0['X'], r1['X'])
    assert 0 < r0['bounds'][1] < g['X_pix'].shape[0]
    events = lambda lines: [ln for ln in lines if 'LR' in ln or 'Reached' in ln]
    assert logs_match(events(r0['log'].tolist()), events(g['log'].tolist()))
    assert r0['E'].shape == g['E'].shape
    assert rel_err(r0['E'][:100], g['E'][:100]) < 1e-8
    assert r0['X'].shape == g['final_X'].shape
    g1 = np.load(os.path.join(GOLDEN, 'f1_tiny_masks.npz'), allow_pickle=False)
    for r in (r0, r1):
        assert r['hist_X'].shape == (5, 60, 3) and r['hist_C'].shape == (5, 15, 3) and r['hist_phi'].shape == (5,)
        assert np.abs(r['hist_X'][:4] - g1['Adam_111__step_X']).max() < 1e-9
        assert np.abs(r['final_X1'] - g1['Adam_111__final_X']).max() < 1e-9
        assert bool(r['refused'])
