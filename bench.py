#!/usr/bin/env python3
"""Headline benchmark: bundle-adjustment observations per second per full iteration.

    python bench.py --gpus N --steps K --warmup W

One "step" is one full iteration of the fit_scene loop on a synthetic video-orbit scene that is
already resident in HBM: fused observation pass (projection, residual, analytic Jacobian,
gradient), sum of the camera-side block over the GPUs, the 8-byte energy read-back the LR schedule
branches on, Adam update of X, C, Theta, phi_x, gauge fix.

Workloads
  N = 1 : 1 M points, 2 k cameras, 10 M observations on one MI355X.
  N > 1 : 10 M points, 2 k cameras, 100 M observations IN TOTAL, split over the N GPUs by point ranges
          (strong scaling; 12.5 M observations per GPU at N = 8), cameras replicated, one sum over the
          ranks per iteration.  The same scene is then timed on ONE GPU (rank 0 builds all shards), so
          that the line carries the 1 -> N speed-up on an identical workload.

With N > 1 and no launcher environment (WORLD_SIZE unset) the script starts its N ranks itself, as
child processes, before it touches the GPU; under torch.distributed.run it is a rank.

Prints ONE JSON line on rank 0: metric/value/unit/... plus `roofline` (gradient pass, the kernel of the
timed loop), `roofline_jtj` (the pass with the J^T J blocks, timed stand-alone) and, at N = 1,
`cpu_baseline` (the C restatement of the reference's iteration on this box's host cores)."""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak
TRAFFIC_FILE = os.path.join('profiles', 'r03_pmc_traffic_cfg3.json')


def measured_traffic(kernel_key, points_per_gpu, cameras, track_len):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE of
    this script's own command, corrected as profiles/summarize_pmc.py documents).  The counters cannot be read
    from inside a run: the figure is a constant of the committed profile, and only for the workload the passes
    were taken on (else None)."""
    path = os.path.join(ROOT, TRAFFIC_FILE)
    if (points_per_gpu, cameras, track_len) != (1_000_000, 2000, 10) or not os.path.exists(path):
        return None
    with open(path) as fh:
        for kernel, row in json.load(fh).items():
            if kernel_key in kernel:
                return row.get('hbm_bytes_per_launch')
    return None


COUNTER_FILE = os.path.join('profiles', 'r03_pmc_run_pass_counters_10M.txt')


def pipe_shares(kernel_key, points_per_gpu, cameras, track_len):
    """How busy the kernel keeps the vector ALU and the LDS (the pass sits near the ridge, so its HBM fraction alone
    says little): shares of the kernel's cycles from the committed SQ counter pass of stand-alone launches on this
    workload - a constant of that profile, like `traffic`.  Kernel cycles = SQ_BUSY_CYCLES / 32 shader engines,
    vector ALU = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs, LDS = SQ_LDS_IDX_ACTIVE / 256 CUs."""
    path = os.path.join(ROOT, COUNTER_FILE)
    if (points_per_gpu, cameras, track_len) != (1_000_000, 2000, 10) or not os.path.exists(path):
        return None
    c = {}
    with open(path) as fh:
        for line in fh:
            if kernel_key in line:
                parts = line.split()
                for k, tok in enumerate(parts):
                    if tok.startswith('SQ_') and k + 1 < len(parts):
                        c[tok] = float(parts[k + 1])
    need = ('SQ_BUSY_CYCLES', 'SQ_ACTIVE_INST_VALU', 'SQ_LDS_IDX_ACTIVE', 'SQ_LDS_BANK_CONFLICT')
    if any(k not in c for k in need) or c['SQ_BUSY_CYCLES'] <= 0:
        return None
    cycles = c['SQ_BUSY_CYCLES'] / 32.0
    return {'valu_busy': c['SQ_ACTIVE_INST_VALU'] * 4.0 / 1024.0 / cycles, 'lds_busy': c['SQ_LDS_IDX_ACTIVE'] / 256.0 / cycles,
            'lds_bank_conflict_share': c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'],
            'source': f'{COUNTER_FILE} (rocprofv3 --pmc pass, committed; not collected by this run)'}


def host_cpu():
    model = '?'
    try:
        with open('/proc/cpuinfo') as fh:
            for line in fh:
                if line.startswith('model name'):
                    model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    return model, os.cpu_count()


def cpu_baseline(sc, budget_s=12.0):
    """Full iterations of the C restatement (oracle/ba_oracle.c) on the host cores: the WHOLE workload of this rank
    on every core the box gives this job (OpenMP), and, beside it, one core on a bounded prefix."""
    from oracle import c_oracle
    model, ncpu = host_cpu()
    threads = max(1, min(16, int(os.environ.get('OMP_NUM_THREADS', '0') or 0) or ncpu or 1))

    def run(pts, n_threads, budget, most):
        cut = int(np.searchsorted(sc['i'], pts, side='left'))
        c_oracle.set_threads(n_threads)
        try:
            it = c_oracle.Iterator(sc['i'][:cut], sc['j'][:cut], sc['uv'][:cut], sc['X_0'][:pts], sc['C_0'], sc['Theta_0'],
                                   sc['phi_x_0'], sc['W'], sc['H'], 1e-3)
            it.step()
            t0 = time.perf_counter()
            n = 0
            while True:
                it.step()
                n += 1
                dt = time.perf_counter() - t0
                if dt > budget or n >= most:
                    break
        finally:
            c_oracle.set_threads(1)
        return cut, n, dt
    cut, n, dt = run(sc['N_local'], threads, budget_s, 200)
    pts1 = min(sc['N_local'], 200_000)
    cut1, n1, dt1 = run(pts1, 1, budget_s / 2, 100)
    return {'value': cut * n / dt, 'unit': 'observations/s', 'cores': threads, 'kind': 'port',
            'host_cpu_model': model, 'host_cpus': ncpu,
            'sample': f'{n} full iterations (evaluate + Adam + gauge fix) of oracle/ba_oracle.c on the whole workload '
                      f'({sc["N_local"]} points / {cut} observations, K={sc["K"]} cameras) with {threads} OpenMP threads of '
                      f'the {ncpu} CPUs the host shows',
            'one_core': {'value': cut1 * n1 / dt1, 'cores': 1,
                         'sample': f'{n1} iterations on the first {pts1} points / {cut1} observations, 1 thread'}}


def grad_pass_bytes(nv, n_pts, K):
    """Algorithmic bytes of the gradient pass: 24 B per observation record (i32, i32, f64, f64), X in, gX out,
    cameras in and out."""
    return 24 * nv + 24 * n_pts + 24 * n_pts + 48 * K + 48 * K


def jtj_pass_bytes(nv, n_pts, K):
    """... of the pass with the J^T J blocks: the same plus V (6 doubles per point) and B, bphi (27 per camera)."""
    return grad_pass_bytes(nv, n_pts, K) + 48 * n_pts + 216 * K


def roofline_block(kernel, algo_bytes, kern_ms, traffic, extra=None):
    avg_ms = float(np.mean(kern_ms)) if len(kern_ms) else float('nan')
    achieved = algo_bytes / (avg_ms * 1e-3) / 1e9
    out = {'bound': 'hbm', 'kernel': kernel, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
           'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
           'traffic_source': (f'{TRAFFIC_FILE}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, '
                              'committed; a constant of that profile, not collected by this run') if traffic else None,
           'achieved_traffic': (traffic / (avg_ms * 1e-3) / 1e9) if traffic else None,
           'frac_physical': (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
           'algorithmic_bytes_per_launch': algo_bytes, 'avg_kernel_ms': avg_ms,
           'median_kernel_ms': float(np.median(kern_ms)) if len(kern_ms) else float('nan'), 'launches_timed': len(kern_ms)}
    if extra:
        out.update(extra)
    return out


def time_jtj_pass(ba, sc, dev, nv_total, launches=24):
    """Kernel time of the pass with the J^T J blocks (stand-alone, HIP events handed to the dispatch) on a
    shard of its own holding the scene `sc`."""
    import torch
    sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], sc['K'], sc['W'], sc['H'], dev,
                         n_visible_total=nv_total)
    try:
        sh.distributed = False          # the kernel alone: no sum over ranks behind it
        sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
        for _ in range(6):
            sh.eval_jtj()
        torch.cuda.synchronize()
        sh.eval_kernel_ms()
        for _ in range(launches):
            sh.eval_jtj(timed=True)
        torch.cuda.synchronize()
        return sh.eval_kernel_ms()
    finally:
        sh.close()


F64_MATRIX_PEAK_TFLOPS = 78.6      # MI355X fp64 matrix rate (= its fp64 vector rate)


def lm_iteration_times(ba, dev, K, N, track_len, iterations=7):
    """Steady-state wall time of whole Levenberg-Marquardt iterations (lm.LMFit: Schur build, Cholesky solve,
    back-substitution, gauge fix, the trial point's J^T J pass, energy read-back) on a scene of N points and K
    cameras, and the device time of the three phases of the damped solve from events around them."""
    import torch
    sc = ba.scenes.orbit_scene(N=N, K=K, track_len=track_len, seed=0)
    sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], K, sc['W'], sc['H'], dev)
    try:
        sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
        sh.lm_marks = []
        fit = ba.lm.LMFit(sh, iterations + 1, lam=1e-3, print_step=10 ** 9, printer=lambda *a: None)
        fit.run()
        torch.cuda.synchronize()
        st = np.diff(np.array(fit.stamps)) * 1e3      # [first evaluation, iteration 1 (sets up S), iteration 2, ...]
        later = st[2:] if len(st) > 2 else st[1:]
        ev = sh.lm_marks
        phases = np.array([[ev[k].elapsed_time(ev[k + 1]) for k in range(q, q + 3)] for q in range(0, len(ev) - 3, 4)])
        ph = phases[1:] if len(phases) > 1 else phases
        E = fit.energies()
        return {'cameras': K, 'points': sc['N_local'], 'observations': int(sc['i'].size), 'reduced_system_unknowns': sh.n_red,
                'iteration_ms': float(later.mean()), 'iteration_ms_min': float(later.min()), 'iterations_timed': int(later.size),
                'first_iteration_ms': float(st[1]), 'schur_build_ms': float(ph[:, 0].mean()),
                'cholesky_solve_ms': float(ph[:, 1].mean()), 'back_substitution_ms': float(ph[:, 2].mean()),
                'accepted': fit.accepted, 'rejected': fit.rejected, 'energy_first': float(E[0]), 'energy_last': float(E[-1])}
    finally:
        sh.close()


def dense_cholesky_rate(ba, dev, n=30001, repeats=2):
    """ba_cholesky_solve on a dense SPD matrix of n unknowns with every tile computed (BA_CHOL_DENSE=1: the default
    path leaves out the structural zeros of a video's reduced camera system, which a dense matrix does not have):
    n^3 / 3 flops over the wall time of the whole call (factorisation + both triangular solves)."""
    import ctypes as C
    import torch
    lib = ba.engine._lib.load()
    ld = (n + 63) // 64 * 64
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.randn(ld, 192, dtype=torch.float64, device=dev, generator=g)
    S0 = A @ A.T
    S0.diagonal().add_(float(n))
    del A
    b0 = torch.randn(ld, dtype=torch.float64, device=dev, generator=g)
    b0[n:] = 0.0
    times, resid = [], None
    os.environ['BA_CHOL_DENSE'] = '1'
    try:
        for rep in range(repeats + 1):
            S, b = S0.clone(), b0.clone()
            info = C.c_int(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ba.engine.check(lib.ba_cholesky_solve(n, ld, C.c_void_p(S.data_ptr()), C.c_void_p(b.data_ptr()), C.byref(info),
                                                  ba.engine._lib.stream_ptr()))
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            if rep == 0:
                r = S0[:n, :n] @ b[:n] - b0[:n]
                resid = float(r.abs().max() / b0[:n].abs().max())
            del S, b
    finally:
        os.environ.pop('BA_CHOL_DENSE', None)
    dt = float(np.mean(times[1:]))
    tflops = n ** 3 / 3.0 / dt / 1e12
    return {'bound': 'mfma_f64', 'kernel': 'ba_cholesky_solve, every tile computed (ba::k_syrk_trailing on v_mfma_f64_4x4x4_4b_f64, '
            'ba::k_panel_rows, ba::k_potrf64)', 'unknowns': n, 'flops': n ** 3 / 3.0, 'ms': dt * 1e3, 'achieved': tflops,
            'peak': F64_MATRIX_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tflops / F64_MATRIX_PEAK_TFLOPS,
            'residual_max_rel': resid, 'calls_timed': repeats}


PREHEAT_ITERATIONS = 700
EVENT_TIMED_LAUNCHES = 48


def preheat(ba, sh, sc, iterations=PREHEAT_ITERATIONS):
    """Keeps the GPU busy with iterations of a throw-away fit (a fixed number: every rank walks the same sequence
    of sums over the ranks; about 50 ms at 10 M observations) before anything is measured: after an
    idle spell the chip raises its clock over the first 20-30 ms of load (measured: the same fit's pass kernel takes
    53 us right after 2 s of idling and 49 us a few hundred iterations later, whatever its iteration number), and
    a short run (W = 5, K = 20 is 2 ms) would otherwise be timed entirely inside that ramp."""
    import torch
    sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    fit = ba.SceneFit(sh, lr=1e-3, max_iters=iterations + 8, patience=10 ** 7, print_step=10 ** 9, printer=lambda *a: None)
    for _ in range(iterations):
        fit.step()
    torch.cuda.synchronize()


def run_loop(ba, sh, sc, steps, warmup, sync, timed=True):
    """`warmup` untimed and `steps` timed full iterations; returns (fit, seconds)."""
    preheat(ba, sh, sc)
    sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    total = steps + warmup
    fit = ba.SceneFit(sh, lr=1e-3, max_iters=total + 8 + EVENT_TIMED_LAUNCHES, patience=1000, print_step=10 ** 9,
                      printer=lambda *a: None)
    for _ in range(warmup):
        if not fit.step():
            raise RuntimeError('the fit stopped during the warm-up')
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        if not fit.step():
            raise RuntimeError('the fit stopped inside the timed region: the iterations counted would not all be full ones')
    sync()
    dt = time.perf_counter() - t0
    # The kernel's own duration for the roofline: HIP events handed to the dispatch of the observation-pass kernel
    # (they hold its begin and end) on the launches of the iterations that FOLLOW the timed region - the same loop in
    # the same state.  Such a launch costs the host about 30 us more than a plain one, so none is inside the region.
    if timed:
        sh.eval_kernel_ms()          # drop anything recorded so far
        fit.timed_every = 1
        for _ in range(EVENT_TIMED_LAUNCHES):
            if not fit.step():
                raise RuntimeError('the fit stopped during the event-timed iterations')
        sync()
    return fit, dt


# ------------------------------------------------------------------------------------------------- #
def launch_ranks(args, argv):
    """N > 1 without a launcher: start the N ranks as child processes (this process never touches the GPU)."""
    import torch
    share = os.environ.get('BA_BENCH_SHARE_GPU') == '1'
    have = torch.cuda.device_count()         # counting devices does not initialise the GPU
    if have < args.gpus and not share:
        print(f'bench.py: --gpus {args.gpus} but this machine shows {have} GPU(s); nothing was run '
              f'(BA_BENCH_SHARE_GPU=1 rehearses the ranks on one GPU)', file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE,
                                      text=True))
    rc = 0
    for p in procs:
        out, _ = p.communicate()
        for line in out.splitlines():      # the one JSON line goes to stdout, anything a library printed to stderr
            print(line, file=sys.stdout if line.startswith('{') else sys.stderr, flush=True)
        rc = max(rc, abs(p.returncode))
    return rc


def exchange_child_env(environ, rank, local_rank, world):
    """Environment of an exchange-trial child: a group of its own on its own port.  Under torch.distributed.run
    the ranks carry the launcher's rendezvous settings (TORCHELASTIC_USE_AGENT_STORE makes every rank a CLIENT of
    the launcher's store, which does not exist on the children's port): those do not travel."""
    port = int(environ.get('MASTER_PORT', '29500')) + 23
    env = dict(environ, MASTER_PORT=str(port), BA_NODE_EXCHANGE='1', RANK=str(rank), LOCAL_RANK=str(local_rank),
               WORLD_SIZE=str(world))
    for name in list(env):
        if name.startswith('TORCHELASTIC_') or name in ('GROUP_RANK', 'ROLE_RANK', 'ROLE_NAME', 'ROLE_WORLD_SIZE',
                                                        'GROUP_WORLD_SIZE', 'LOCAL_WORLD_SIZE'):
            del env[name]
    return env


def exchange_trial(args, rank, world, local_rank):
    """The same distributed loop with the sums over the ranks carried by the library's node exchange instead of
    RCCL, in CHILD processes of the ranks (a group of their own on another port): whatever happens to them, the
    measurement already taken stands.  Returns the child's report on rank 0 (a dict) or an error string."""
    env = exchange_child_env(os.environ, rank, local_rank, world)
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', str(world), '--steps', str(args.steps), '--warmup',
           str(args.warmup), '--cameras', str(args.cameras), '--track-len', str(args.track_len), '--total-points',
           str(args.total_points), '--exchange-child']
    try:
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired:
        return 'no result within 240 s'
    if res.returncode != 0:
        tail = (res.stderr or '').strip().splitlines()[-1:] or ['']
        return f'exit status {res.returncode}: {tail[0][:200]}'
    for line in res.stdout.splitlines():
        if line.startswith('{'):
            return json.loads(line)
    return 'no report' if rank == 0 else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--warmup', type=int, default=300)
    ap.add_argument('--points-per-gpu', type=int, default=1_000_000, help='N = 1: points of the scene')
    ap.add_argument('--total-points', type=int, default=10_000_000, help='N > 1: points of the scene, split over the GPUs')
    ap.add_argument('--cameras', type=int, default=2000)
    ap.add_argument('--track-len', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-jtj', action='store_true', help='skip the stand-alone timing of the J^T J pass')
    ap.add_argument('--no-lm', action='store_true', help='N = 1: skip the second-order measurements (roofline_lm)')
    ap.add_argument('--no-one-gpu-reference', action='store_true',
                    help='N > 1: skip the timing of the whole scene on one GPU')
    ap.add_argument('--exchange-trial', action='store_true',
                    help='N > 1: a second measurement with the node exchange kernel as the carrier, in child processes '
                         '(off by default: that kernel has never run across xGMI, and the measurement that counts must '
                         'not share a machine with its first attempt)')
    ap.add_argument('--exchange-child', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}', file=sys.stderr)
        sys.exit(2)
    share = os.environ.get('BA_BENCH_SHARE_GPU') == '1'       # rehearsal: every rank on GPU 0, sums through gloo
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        if share:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)

    t_start = time.perf_counter()

    def note(msg):      # progress on stderr: a run over 100 M observations is silent for a minute at a time otherwise
        if rank == 0 and not args.exchange_child:
            print(f'[bench.py {time.perf_counter() - t_start:6.1f} s] {msg}', file=sys.stderr, flush=True)

    ba = importlib.import_module('3d_reconstruction_amd')
    N_total = args.points_per_gpu if world == 1 else args.total_points
    note(f'building the scene ({N_total} points, part {rank + 1} of {world})')
    sc = ba.scenes.orbit_scene(N=N_total, K=args.cameras, track_len=args.track_len, seed=0, part=rank, parts=world)
    nv_local = int(sc['i'].size)
    nv_t = torch.tensor([nv_local], dtype=torch.int64, device='cpu' if share else dev)
    if world > 1:
        dist.all_reduce(nv_t)
    nv_total = int(nv_t.item())

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device='cpu' if share else dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    note(f'{nv_total} observations in all, {nv_local} here: problem set-up, pre-heating, timed loop')
    # set-up of this rank's share, timed with the list already on the device: the connectivity check of the
    # reference (p2w:519-525) on the device, then point order + tiles + record layouts (engine.Shard)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    i_d, j_d, uv_d = (torch.from_numpy(sc[k]).to(dev) for k in ('i', 'j', 'uv'))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    components = ba.pix2world.visibility_graph_components(i_d, j_d, sc['N_local'], sc['K'], dev)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    sh = ba.engine.Shard(i_d, j_d, uv_d, sc['N_local'], sc['K'], sc['W'], sc['H'], dev, n_visible_total=nv_total)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    setup = {'upload_ms': (t1 - t0) * 1e3, 'connectivity_ms': (t2 - t1) * 1e3, 'components': components,
             'order_tiles_records_ms': (t3 - t2) * 1e3,
             'note': 'one-off, outside the timed loop; connectivity and the point order run on the device copy of the list'}
    del i_d, j_d, uv_d
    fit, dt = run_loop(ba, sh, sc, args.steps, args.warmup, sync)
    dt = max_over_ranks(dt)
    kern_ms = sh.eval_kernel_ms()
    total = args.steps + args.warmup
    E_first, E_last = float(fit.E[args.warmup]), float(fit.E[total - 1])
    path = sh.rank_sum_path
    fused = bool(fit.fuse)

    if args.exchange_child:
        if rank == 0:
            print(json.dumps({'ms_per_step': dt / args.steps * 1e3, 'sum_over_ranks': path,
                              'energy_last_timed': E_last}), flush=True)
        sh.close()
        sync()
        dist.destroy_process_group()
        return

    N_loc, K = sc['N_local'], sc['K']
    out = None
    if rank == 0:
        if world == 1:
            workload = (f'synthetic video-orbit scene, {N_total} points, K={K} cameras, track length {args.track_len}: '
                        f'|V|={nv_total} observations on one GPU; full fit_scene iteration (fused observation pass + '
                        f'Adam + gauge fix), gradient-only pass')
        else:
            workload = (f'synthetic video-orbit scene, {N_total} points, K={K} cameras, track length {args.track_len}: '
                        f'|V|={nv_total} observations in total, split by point ranges over {world} GPUs '
                        f'({nv_local} on rank 0), cameras replicated; full fit_scene iteration (fused observation '
                        f'pass + sum of the camera block over the ranks + Adam + gauge fix), gradient-only pass')
        out = {
            'metric': 'BA observations/sec per iteration', 'value': nv_total * args.steps / dt,
            'unit': 'observations/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak' if world == 1 else 'strong',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': workload, 'points': N_total, 'cameras': K, 'observations': nv_total,
                       'preheat_iterations': PREHEAT_ITERATIONS,
                       'parallelism': f'point-sharded x{world}, cameras replicated',
                       'sum_over_ranks': path, 'energy_first_timed': E_first, 'energy_last_timed': E_last,
                       'setup_ms': setup},
            'roofline': roofline_block(
                'ba::k_run_pass<ba::GradOp, ba::RecI16> (fused observation pass: gradient + point update)',
                grad_pass_bytes(nv_local, N_loc, K), kern_ms,
                measured_traffic('k_run_pass<ba::GradOp', N_loc, K, args.track_len) if fused else None,
                {'observations_per_launch': nv_local,
                 'pipes': pipe_shares('k_run_pass<ba::GradOp', N_loc, K, args.track_len),
                 'note': 'achieved = algorithmic bytes (24 B per observation record + X in + gX out + cameras in '
                         'and out) / kernel time; the kernel streams 8-byte records and carries the fused point '
                         'update, so the bytes it really moves are `traffic`'}),
        }
        # what an iteration spends outside the observation pass: the camera step, the hand-over between the two
        # launches and, with several ranks, the sum over the ranks - the part that does not shrink with the shard
        if len(kern_ms):
            out['serial_us'] = (dt / args.steps - float(np.mean(kern_ms)) * 1e-3) * 1e6
    sh.close()
    del sh, fit

    # ---- the pass with the J^T J blocks, stand-alone, on this rank's shard (and, at N = 1, on the 12.5 M-observation
    # share a GPU holds when 100 M observations are split eight ways)
    if not args.no_jtj:
        note(f'timed loop done ({dt / args.steps * 1e6:.1f} us per iteration); the pass with the J^T J blocks')
        jt = time_jtj_pass(ba, sc, dev, nv_total)
        if rank == 0:
            out['roofline_jtj'] = roofline_block(
                'ba::k_run_pass<ba::JtjOp, ba::RecI16> (fused pass: residual + Jacobian + gradient + J^T J blocks)',
                jtj_pass_bytes(nv_local, N_loc, K), jt,
                measured_traffic('k_run_pass<ba::JtjOp', N_loc, K, args.track_len),
                {'observations_per_launch': nv_local, 'timed': 'stand-alone launches, outside the timed loop',
                 'pipes': pipe_shares('k_run_pass<ba::JtjOp', N_loc, K, args.track_len),
                 'parity': 'second-order extension without a counterpart in the reference: its blocks are pinned by '
                           "the reference's own per-observation Jacobians (golden F11)"})
        if world == 1:
            sc8 = ba.scenes.orbit_scene(N=args.total_points, K=args.cameras, track_len=args.track_len, seed=0, part=0,
                                        parts=8)
            jt8 = time_jtj_pass(ba, sc8, dev, int(sc8['i'].size))
            out['roofline_jtj']['at_one_eighth_of_100M'] = {
                'observations_per_launch': int(sc8['i'].size),
                'avg_kernel_ms': float(np.mean(jt8)),
                'frac': jtj_pass_bytes(int(sc8['i'].size), sc8['N_local'], K) / (float(np.mean(jt8)) * 1e-3) / 1e9
                / HBM_PEAK_GBS}
            del sc8

    # ---- N > 1: the same scene on ONE GPU (rank 0 builds every rank's shard), for the 1 -> N speed-up
    if world > 1 and not args.no_one_gpu_reference:
        note('the whole scene on one GPU (rank 0 builds every part; the other ranks wait)')
        if rank == 0:
            # rank 0 builds the other ranks' shards of the scene itself (seconds each; the others wait at the barrier)
            parts = [sc] + [ba.scenes.orbit_scene(N=N_total, K=args.cameras, track_len=args.track_len, seed=0, part=r,
                                                  parts=world) for r in range(1, world)]
            offs = np.cumsum([0] + [g['N_local'] for g in parts])
            whole = dict(sc, i=np.concatenate([(g['i'].astype(np.int64) + offs[r]).astype(np.int32)
                                               for r, g in enumerate(parts)]),
                         j=np.concatenate([g['j'] for g in parts]), uv=np.concatenate([g['uv'] for g in parts]),
                         X_0=np.concatenate([g['X_0'] for g in parts]), N_local=int(offs[-1]))
            del parts
            one = ba.engine.Shard(whole['i'], whole['j'], whole['uv'], whole['N_local'], K, sc['W'], sc['H'], dev)
            one.distributed = False
            steps1 = max(10, min(args.steps, 100))
            fit1, dt1 = run_loop(ba, one, whole, steps1, min(args.warmup, 100), torch.cuda.synchronize, timed=False)
            out['strong_scaling'] = {
                'one_gpu_ms_per_step': dt1 / steps1 * 1e3, 'one_gpu_value': nv_total * steps1 / dt1,
                'speedup': (dt1 / steps1) / (dt / args.steps),
                'note': f'the identical scene ({nv_total} observations) timed on rank 0 alone, {steps1} iterations'}
            one.close()
            del one, whole, fit1
        dist.barrier()

    # ---- N > 1: the node exchange kernel as the carrier of the sums, measured by child processes
    if world > 1 and args.exchange_trial:
        note('the same loop with the node exchange kernel, in child processes')
        rep = exchange_trial(args, rank, world, local_rank)
        if rank == 0:
            out['config']['transports'] = {
                path: {'ms_per_step': out['ms_per_step']},
                'node exchange kernel (opt-in, BA_NODE_EXCHANGE=1)': rep if isinstance(rep, dict) else {'error': rep}}
        dist.barrier()

    # ---- N = 1: the second-order path - whole LM iterations at 2 k and at 20 k cameras, and the dense fp64 solve
    if world == 1 and not args.no_lm:
        note('Levenberg-Marquardt iterations at K = 2000 and K = 20000, dense Cholesky at n = 30001')
        torch.cuda.empty_cache()
        out['roofline_lm'] = {
            'cfg3': lm_iteration_times(ba, dev, args.cameras, args.points_per_gpu, args.track_len),
            'cfg5': lm_iteration_times(ba, dev, 20000, args.points_per_gpu, args.track_len),
            'dense_solve': dense_cholesky_rate(ba, dev),
            'note': 'iteration_ms: host clock between consecutive energy read-backs of lm.LMFit in the steady state; the '
                    'phase times are device time between events around Shard.lm_step\'s three calls'}
        torch.cuda.empty_cache()

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            note('CPU baseline (one host core)')
            out['cpu_baseline'] = cpu_baseline(sc)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
