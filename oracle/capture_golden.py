"""Generate the golden vectors under tests/golden/ by running the reference itself.

Runs ONLY in the build container (it imports /root/reference/modules/pix2world.py, which
never travels to the GPU box).  What is committed is data: inputs and the reference's
outputs, as .npz / .json.  Nothing of the reference's source is written anywhere.

    MPLBACKEND=Agg python oracle/capture_golden.py [--out tests/golden]

Capture hooks (no reference source is touched):
  * `gd_func_dict[optimizer]` is looked up by name on every step (p2w:705): a recording
    wrapper placed in that registry sees (parameters, gradient) of every group in the order
    X, C, Theta, phi_x and forwards to the original update rule;
  * the module-level name `lr_manager` (called at p2w:617) is wrapped to record the iteration
    counter on entry/exit and whether a rollback happened;
  * stdout is captured for the LR-event / stop-reason lines (p2w:405, 430, 626-641).
"""
from __future__ import annotations

import argparse
import contextlib
import io
import json
import os
import pickletools
import sys

import numpy as np

REF_DIR = '/root/reference'
sys.path.insert(0, os.path.join(REF_DIR, 'modules'))
os.environ.setdefault('MPLBACKEND', 'Agg')
import pix2world as p  # noqa: E402  (the reference)


# --------------------------------------------------------------------------- #
def make_scene(N, K, seed, vis_frac=1.0, W=1920, H=1080, phi_x=1.2):
    """Synthetic scene from the reference's own generators (p2w:2338-2427) plus a visibility
    drop mask and a noisy initial guess."""
    C, R = p.generate_cameras(K, r_min=4, r_max=6, theta_min=np.pi / 3, theta_max=2 * np.pi / 3,
                              seed=seed)
    Theta = p.get_rotations(R)
    AR, ARC = p.form_view_matrices(C, R, W=W, H=H, phi_x=phi_x)
    vb = p.find_bounding_box(AR.reshape(-1, 3), ARC.ravel())
    X = p.generate_points(N, AR, ARC, vb, np.array([[-10, 10]] * 3, dtype=float), seed=seed + 1)
    model = p.transform(X, C, Theta, phi_x, W, H, delete_boundary_points=True)
    X_pix = np.where(np.isnan(model), -1, np.floor(model)).astype(np.int32)
    if vis_frac < 1.0:
        drop = np.random.default_rng(seed + 2).random((N, K)) > vis_frac
        X_pix[drop] = -1
    rng = np.random.default_rng(5)
    X0 = X + rng.normal(0, 0.05, X.shape)
    C0 = C + rng.normal(0, 0.05, C.shape)
    Th0 = Theta + rng.normal(0, 0.01, Theta.shape)
    return dict(X_pix=X_pix, W=W, H=H, X_true=X, C_true=C, Theta_true=Theta, phi_true=phi_x,
                X_0=X0, C_0=C0, Theta_0=Th0, phi_x_0=phi_x * 1.02)


def run_reference(scene, lr, max_iters, optimizer='Adam', masks=(True, True, True), frame_hook=None, **kw):
    """Run p.fit_scene with the recording hooks installed; returns a dict of arrays.
    `frame_hook(step_index, locals_of_fit_scene)` is called once per update, before the first
    group's update rule runs: the registry entry is called from inside fit_scene (p2w:705), so the
    caller's frame holds that iteration's intermediate arrays (read only, nothing is modified)."""
    steps = []          # one entry per executed update: {'X':(x,g), 'C':..., ...}
    cur = {}
    names = []
    if masks[0]:
        names.append('X')
    if masks[1]:
        names.extend(['C', 'Theta'])
    if masks[2]:
        names.append('phi_x')
    orig_update = p.gd_func_dict[optimizer]
    orig_manager = p.lr_manager
    manager_calls = []

    def rec_update(x, grad, lr_, **state):
        if frame_hook is not None and not cur:
            frame_hook(len(steps), sys._getframe(1).f_locals)
        name = names[len(cur)]
        cur[name] = (np.array(x, dtype=np.float64).copy(), np.array(grad, dtype=np.float64).copy())
        if len(cur) == len(names):
            steps.append(dict(cur))
            cur.clear()
        return orig_update(x, grad, lr_, **state)

    def rec_manager(lr_, iters, *a, **k):
        res = orig_manager(lr_, iters, *a, **k)
        manager_calls.append((iters, res[0][1], len(res[1]), float(res[0][0])))
        return res

    p.gd_func_dict[optimizer] = rec_update
    p.lr_manager = rec_manager
    out = io.StringIO()
    try:
        with contextlib.redirect_stdout(out):
            ret = p.fit_scene(scene['X_pix'], scene['W'], scene['H'], lr, max_iters,
                              scene['X_0'], scene['C_0'], scene['Theta_0'], scene['phi_x_0'],
                              X_mask=masks[0], C_Theta_mask=masks[1], phi_x_mask=masks[2],
                              optimizer=optimizer, **kw)
    finally:
        p.gd_func_dict[optimizer] = orig_update
        p.lr_manager = orig_manager
    res = {'E': np.asarray(ret[len(names)]), 'log': out.getvalue().splitlines(),
           'manager_calls': np.array(manager_calls, dtype=np.float64).reshape(-1, 4)}
    for n, v in zip(names, ret[:len(names)]):
        res['final_' + n] = np.asarray(v, dtype=np.float64)
    for n in names:
        res['step_' + n] = np.array([s[n][0] for s in steps])
        res['grad_' + n] = np.array([s[n][1] for s in steps])
    return res


def save_npz(path, **arrays):
    clean = {}
    for k, v in arrays.items():
        if isinstance(v, (list, tuple)) and v and isinstance(v[0], str):
            v = np.array(v)
        clean[k] = v
    np.savez_compressed(path, **clean)
    print('wrote', path, f'{os.path.getsize(path) / 1024:.1f} KiB')


# --------------------------------------------------------------------------- #
def capture_f1(out):
    """Tiny dense-API scene, every mask combination x {Adam, SGD}, 4 iterations."""
    sc = make_scene(60, 15, seed=3, vis_frac=0.6)
    arrays = {k: sc[k] for k in ('X_pix', 'X_0', 'C_0', 'Theta_0')}
    arrays['W'], arrays['H'], arrays['phi_x_0'] = sc['W'], sc['H'], sc['phi_x_0']
    combos = [(a, b, c) for a in (True, False) for b in (True, False) for c in (True, False)
              if a or b or c]
    for opt in ('Adam', 'SGD'):
        lr = 1e-3 if opt == 'Adam' else 1e-7
        for m in combos:
            tag = f"{opt}_{''.join('1' if t else '0' for t in m)}"
            r = run_reference(sc, lr=lr, max_iters=5, optimizer=opt, masks=m)
            for k, v in r.items():
                if k != 'log':
                    arrays[f'{tag}__{k}'] = v
            arrays[f'{tag}__log'] = np.array(r['log'])
    arrays['combos'] = np.array(combos)
    save_npz(os.path.join(out, 'f1_tiny_masks.npz'), **arrays)


def moscow_init(X_pix, W, H):
    """Deterministic initial guess for the moscow_ride tracks: cameras on a line along z,
    identity rotations, phi_x = 1.7, every point on the ray through its first visible pixel
    at depth 5 (direction from the reference's reverse_project, p2w:164-189)."""
    N, K = X_pix.shape[:2]
    C0 = np.zeros((K, 3))
    C0[:, 2] = 0.01 * np.arange(K)
    Th0 = np.zeros((K, 3))
    phi0 = 1.7
    rays = p.reverse_project(X_pix, C0, Th0, phi0, W, H, dir_only=True)
    first = np.argmax(X_pix[:, :, 0] >= 0, axis=1)
    X0 = C0[first] + 5.0 * rays[np.arange(N), first]
    return X0, C0, Th0, phi0


def read_tracks_pickle(path):
    """Read the demo's `(X_pix ndarray, W, H)` pickle WITHOUT unpickling it: the opcode stream
    is only disassembled (pickletools.genops executes nothing) and the array is rebuilt from
    its shape tuple, dtype string and raw byte payload."""
    with open(path, 'rb') as fh:
        data = fh.read()
    ints, strs, payload, shape = [], [], None, None
    for op, arg, _ in pickletools.genops(data):
        if op.name in ('BININT', 'BININT1', 'BININT2', 'LONG1'):
            ints.append(int(arg))
        elif op.name in ('SHORT_BINUNICODE', 'BINUNICODE'):
            strs.append(arg)
        elif op.name in ('BINBYTES', 'BINBYTES8') and payload is None:
            payload = arg
            after_payload = len(ints)
        elif op.name == 'TUPLE3' and shape is None and len(ints) >= 4:
            shape = tuple(ints[-3:])       # first 3-tuple of ints after the class triple
    allowed = {'numpy.core.multiarray', 'numpy._core.multiarray', '_reconstruct', 'numpy',
               'ndarray', 'dtype', 'i4', '<'}
    if payload is None or shape is None or not set(strs) <= allowed or 'i4' not in strs:
        raise ValueError('unexpected pickle layout in ' + path)
    X_pix = np.frombuffer(payload, dtype='<i4').reshape(shape).copy()
    W, H = ints[after_payload:after_payload + 2]
    return X_pix, W, H


def capture_f2(out):
    """moscow_ride demo tracks (the reference's only real data set)."""
    X_pix, W, H = read_tracks_pickle(os.path.join(REF_DIR, 'demo_examples/moscow_ride/input.txt'))
    X0, C0, Th0, phi0 = moscow_init(X_pix, W, H)
    sc = dict(X_pix=X_pix, W=W, H=H, X_0=X0, C_0=C0, Theta_0=Th0, phi_x_0=phi0)
    vis = X_pix[:, :, 0] >= 0
    ii, jj = np.nonzero(vis)
    short = run_reference(sc, lr=1e-3, max_iters=51, patience=1000)
    # Free run: the ray initialisation puts every point exactly on the ray of its first view, so
    # some residuals (and with them camera 0's whole gradient) are rounding noise at iteration 0,
    # and Adam's first step g / (1e-12 + |g|) turns that noise into O(lr) moves.  Any change of
    # summation order then changes the trajectory.  The free run therefore starts from the same
    # initialisation with the points moved off their rays by a fixed pseudo-random offset.
    X0_free = X0 + np.random.default_rng(17).normal(0.0, 0.02, X0.shape)
    long_ = run_reference(dict(sc, X_0=X0_free), lr=1e-3, max_iters=300, patience=1000)
    arrays = dict(N=X_pix.shape[0], K=X_pix.shape[1], W=W, H=H,
                  obs_i=ii.astype(np.int32), obs_j=jj.astype(np.int32),
                  obs_xy=X_pix[vis].astype(np.int32),
                  X_0=X0, C_0=C0, Theta_0=Th0, phi_x_0=phi0)
    for k, v in short.items():
        arrays['tf__' + k] = np.array(v) if k == 'log' else v
    arrays['free__X_0'] = X0_free
    for k in ('E', 'final_X', 'final_C', 'final_Theta', 'final_phi_x', 'manager_calls'):
        arrays['free__' + k] = long_[k]
    arrays['free__log'] = np.array(long_['log'])
    save_npz(os.path.join(out, 'f2_moscow_ride.npz'), **arrays)


def capture_f3(out):
    """LR-event run: small patience so that increases / decreases / rollbacks all fire."""
    sc = make_scene(50, 12, seed=7, vis_frac=1.0)
    r = run_reference(sc, lr=1e-3, max_iters=200, patience=20)
    arrays = {k: sc[k] for k in ('X_pix', 'X_0', 'C_0', 'Theta_0')}
    arrays['W'], arrays['H'], arrays['phi_x_0'] = sc['W'], sc['H'], sc['phi_x_0']
    for k, v in r.items():
        arrays[k] = np.array(v) if k == 'log' else v
    save_npz(os.path.join(out, 'f3_lr_events.npz'), **arrays)


def capture_f4(out):
    """lr_manager on scripted energy sequences with a counting recompute function."""
    rng = np.random.default_rng(11)
    cases = []
    for case in range(40):
        n = int(rng.integers(5, 60))
        patience = int(rng.integers(0, 5))
        factor = float(rng.choice([2.0, 3.0, 1.5]))
        kind = case % 4
        if kind == 0:
            E = np.cumsum(rng.normal(-0.1, 1.0, n)) + 50.0
        elif kind == 1:
            E = 100.0 * np.exp(-0.1 * np.arange(n)) * (1 + 0.3 * rng.random(n))
        elif kind == 2:
            E = rng.integers(0, 6, n).astype(float)          # many exact ties
        else:
            E = 10.0 + np.abs(rng.normal(0, 1, n)).cumsum()  # never improves
        params = {'a': {'main': np.zeros(3)}, 'b': {'main': np.zeros(1)}}
        p.form_params_dict(params, 'Adam', True, 4 * n + 8)
        calls = [0]

        def func(tag, calls=calls):
            calls[0] += 1
            return (tag,), {}

        lr, iters = 1.0, 0
        E_arr = np.zeros(4 * n + 8)
        E_min = E_min_temp = np.inf
        pt, st, began, inc = 0, patience, False, False
        trace = []
        out_io = io.StringIO()
        with contextlib.redirect_stdout(out_io):
            for step in range(n):
                E_arr[iters] = E[step]
                # emulate the optimiser touching the state between calls
                for val in params.values():
                    val['main'] += 1.0
                    val['gd']['s'] += 0.5
                    val['gd']['t'] += 1
                tup, lst = p.lr_manager(lr, iters, E_arr, E_min, E_min_temp, pt, st, began, inc,
                                        patience, factor, params, True, func, ('x',), {})
                lr, iters, E_min, E_min_temp, pt, st, began, inc = tup
                trace.append(dict(lr=float(lr), iters=int(iters), E_min=float(E_min),
                                  E_min_temp=float(E_min_temp), patience_timer=int(pt),
                                  success_timer=int(st), began=bool(began), increased=bool(inc),
                                  recomputed=len(lst),
                                  main_a=params['a']['main'].tolist(),
                                  release_a=params['a']['release'].tolist(),
                                  s_a=params['a']['gd']['s'].tolist(),
                                  t_a=int(params['a']['gd']['t'][0])))
                iters += 1
        cases.append(dict(E=E.tolist(), patience=patience, factor=factor, trace=trace,
                          log=out_io.getvalue().splitlines(), recompute_calls=calls[0]))
    path = os.path.join(out, 'f4_lr_manager.json')
    with open(path, 'w') as fh:
        json.dump(cases, fh)
    print('wrote', path, f'{os.path.getsize(path) / 1024:.1f} KiB')


def capture_f5(out):
    """Gauge / rotation / optimiser / scoring functions called stand-alone."""
    rng = np.random.default_rng(21)
    K, N = 9, 14
    Theta = rng.uniform(-1.2, 1.2, (K, 3))
    X = rng.normal(0, 3, (N, 3))
    C = rng.normal(0, 2, (K, 3))
    R = p.ryxz(Theta)
    Xa, Ca, Tha, Ra = p.align_scene(X, C, Theta=Theta, ret_R=True)
    Xn, Cn, j1 = p.normalize_scene(Xa, Ca, ret_j=True)
    x = rng.normal(0, 1, (N, 3))
    g = rng.normal(0, 1, (N, 3))
    s = rng.normal(0, 0.1, (N, 3))
    r = rng.random((N, 3))
    xa, st = p.gd_adam(x, g, 1e-2, s, r, np.array([3]))
    xs, _ = p.sgd(x, g, 1e-2)
    # transform / error on a scene with points behind cameras and outside the frame
    W, H, phi = 640, 480, 1.1
    Xt = rng.normal(0, 4, (N, 3))
    Ct = rng.normal(0, 1, (K, 3))
    T1 = p.transform(Xt, Ct, Theta, phi, W, H)
    T2 = p.transform(Xt, Ct, Theta, phi, W, H, delete_boundary_points=True)
    T3 = p.transform(Xt, Ct, Theta, phi, W, H, delete_back_points=False)
    Xp = np.where(np.isnan(T2), -1, np.floor(T2)).astype(np.int32)
    Xp[rng.random((N, K)) > 0.8] = -1
    err1 = p.error(Xp, T1)
    err2 = p.error(Xp, p.transform(Xt + 0.01, Ct, Theta, phi, W, H))
    save_npz(os.path.join(out, 'f5_gauge_funcs.npz'),
             Theta=Theta, X=X, C=C, R=R, Rx=p.rx(Theta[:, 0]), Ry=p.ry(Theta[:, 1]),
             Rz=p.rz(Theta[:, 2]), Theta_back=p.get_rotations(R),
             align_X=Xa, align_C=Ca, align_Theta=Tha, align_R=Ra,
             norm_X=Xn, norm_C=Cn, norm_j1=j1,
             adam_x=x, adam_g=g, adam_s=s, adam_r=r, adam_x_new=xa, adam_s_new=st['s'],
             adam_r_new=st['r'], adam_t_new=st['t'], sgd_x_new=xs,
             tr_X=Xt, tr_C=Ct, tr_W=W, tr_H=H, tr_phi=phi, tr_default=T1, tr_boundary=T2,
             tr_keep_back=T3, err_X_pix=Xp, err_value=err1, err_value_shifted=err2)


def capture_f6(out):
    """Medium scene, 500 free-running iterations with LR events (patience 100)."""
    sc = make_scene(2000, 40, seed=13, vis_frac=0.5)
    r = run_reference(sc, lr=1e-3, max_iters=500, patience=100)
    vis = sc['X_pix'][:, :, 0] >= 0
    ii, jj = np.nonzero(vis)
    keep = [0, 1, 2, 3, 50, 150, 300, 498]
    arrays = dict(N=2000, K=40, W=sc['W'], H=sc['H'], obs_i=ii.astype(np.int32),
                  obs_j=jj.astype(np.int32), obs_xy=sc['X_pix'][vis].astype(np.int32),
                  X_0=sc['X_0'], C_0=sc['C_0'], Theta_0=sc['Theta_0'], phi_x_0=sc['phi_x_0'],
                  E=r['E'], log=np.array(r['log']), manager_calls=r['manager_calls'],
                  final_X=r['final_X'], final_C=r['final_C'], final_Theta=r['final_Theta'],
                  final_phi_x=r['final_phi_x'], keep=np.array(keep))
    for n in ('X', 'C', 'Theta', 'phi_x'):
        arrays['step_' + n] = r['step_' + n][keep]
        arrays['grad_' + n] = r['grad_' + n][keep]
    save_npz(os.path.join(out, 'f6_medium.npz'), **arrays)


def capture_f7(out):
    """Linear initialisation steps (p2w:1655-1808): triangulation from cameras and resection from
    points on a sparse scene, with the true parameters and with perturbed ones, both pixel
    normalisations.  Also a scene with degenerate points / cameras (too few observations)."""
    import warnings
    sc = make_scene(120, 24, seed=21, vis_frac=0.55)
    X_pix = sc['X_pix'].copy()
    X_pix[:3, 1:] = -1            # points seen once
    X_pix[3, :] = -1              # a point never seen
    X_pix[10:, 5] = -1            # a camera with fewer than 6 points (it keeps what points 0..9 give it)
    vis = X_pix[:, :, 0] >= 0
    ii, jj = np.nonzero(vis)
    arrays = dict(N=120, K=24, W=sc['W'], H=sc['H'], phi_x=sc['phi_true'], obs_i=ii.astype(np.int32),
                  obs_j=jj.astype(np.int32), obs_xy=X_pix[vis].astype(np.int32), X_true=sc['X_true'],
                  C_true=sc['C_true'], Theta_true=sc['Theta_true'], X_noisy=sc['X_0'], C_noisy=sc['C_0'],
                  Theta_noisy=sc['Theta_0'])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for tag, (Xs, Cs, Ts) in dict(true=(sc['X_true'], sc['C_true'], sc['Theta_true']),
                                      noisy=(sc['X_0'], sc['C_0'], sc['Theta_0'])).items():
            for norm in (False, True):
                n = 'norm' if norm else 'pix'
                arrays[f'tri_{tag}_{n}'] = p.get_X_from_C_Theta(X_pix, Cs, Ts, sc['phi_true'], sc['W'], sc['H'],
                                                              normalize=norm)
                Cr, Tr = p.get_C_Theta_from_X(X_pix, Xs, sc['phi_true'], sc['W'], sc['H'], normalize=norm)
                arrays[f'res_C_{tag}_{n}'], arrays[f'res_Theta_{tag}_{n}'] = Cr, Tr
        # points with NaN coordinates (what a failed triangulation hands on) are skipped by the resection
        Xn = sc['X_0'].copy()
        Xn[20:30] = np.nan
        Cr, Tr = p.get_C_Theta_from_X(X_pix, Xn, sc['phi_true'], sc['W'], sc['H'], normalize=True, _validate=False)
        arrays['X_with_nan'], arrays['res_C_nan_norm'], arrays['res_Theta_nan_norm'] = Xn, Cr, Tr
    save_npz(os.path.join(out, 'f7_linear_init.npz'), **arrays)


def make_subscenes(N, P, seed, present_frac=0.7, noise=0.01):
    """P noisy copies of one point cloud, each in its own similarity frame, with missing entries
    (the input of p2w:729: X (N, P, 3), NaN where a subscene does not hold a point)."""
    rng = np.random.default_rng(seed)
    X_true = rng.normal(0, 2.0, (N, 3))
    S = np.exp(rng.normal(0, 0.3, P))
    Psi = rng.normal(0, 0.4, (P, 3))
    T = rng.normal(0, 1.0, (P, 3))
    S[0], Psi[0], T[0] = 1.0, 0.0, 0.0
    R = p.ryxz(Psi)
    # subscene j holds x_ij with  S_j R_j x_ij + T_j = X_i (+ noise)
    X = np.einsum('jlk,ijl->ijk', R, (X_true[:, None, :] - T[None]) / S[None, :, None])
    X = X + rng.normal(0, noise, X.shape)
    X[rng.random((N, P)) > present_frac] = np.nan
    return X, S, T, Psi


def run_reference_subscenes(X, lr, max_iters, optimizer='Adam', **kw):
    """p.fit_subscenes_GD with the recording hook in the optimiser registry (groups S, T, Psi)."""
    names, cur, steps = ['S', 'T', 'Psi'], {}, []
    orig_update = p.gd_func_dict[optimizer]

    def rec_update(x, grad, lr_, **state):
        name = names[len(cur)]
        cur[name] = (np.array(x, dtype=np.float64).copy(), np.array(grad, dtype=np.float64).copy())
        if len(cur) == len(names):
            steps.append(dict(cur))
            cur.clear()
        return orig_update(x, grad, lr_, **state)

    p.gd_func_dict[optimizer] = rec_update
    out = io.StringIO()
    try:
        with contextlib.redirect_stdout(out):
            ret = p.fit_subscenes_GD(X, lr, max_iters, optimizer=optimizer, **kw)
    finally:
        p.gd_func_dict[optimizer] = orig_update
    res = {'X_new': ret[0], 'final_S': ret[1], 'final_T': ret[2], 'final_Psi': ret[3], 'E': ret[4],
           'log': out.getvalue().splitlines()}
    for n in names:
        res['step_' + n] = np.array([s[n][0] for s in steps])
        res['grad_' + n] = np.array([s[n][1] for s in steps])
    return res


def capture_f8(out):
    """fit_subscenes_GD (p2w:729-949): Adam and SGD from the default start, Adam from a given start,
    and a run with a short patience so that the schedule raises and lowers the learning rate (cut at 56
    iterations: a little later the run sits at its optimum, the schedule compares energies that are
    equal up to rounding, and its decisions stop being reproducible by any other summation order)."""
    import warnings
    X, S, T, Psi = make_subscenes(90, 6, seed=31)
    arrays = dict(X=X, S_true=S, T_true=T, Psi_true=Psi)
    rng = np.random.default_rng(32)
    S0 = S * np.exp(rng.normal(0, 0.05, S.shape))
    T0 = T + rng.normal(0, 0.1, T.shape)
    P0 = Psi + rng.normal(0, 0.05, Psi.shape)
    arrays.update(S_0=S0, T_0=T0, Psi_0=P0)
    runs = dict(adam=dict(lr=1e-2, max_iters=40, optimizer='Adam'),
                sgd=dict(lr=1e-2, max_iters=40, optimizer='SGD'),
                start=dict(lr=1e-3, max_iters=300, optimizer='Adam', S_0=S0, T_0=T0, Psi_0=P0),
                events=dict(lr=2e-3, max_iters=56, optimizer='Adam', S_0=S0, T_0=T0, Psi_0=P0, patience=12))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for tag, kw in runs.items():
            kw = dict(kw)
            res = run_reference_subscenes(X, kw.pop('lr'), kw.pop('max_iters'), **kw)
            for k, v in res.items():
                arrays[f'{tag}_{k}'] = v
            print(tag, 'E0', res['E'][0], 'E_last', res['E'][-1], 'log lines', len(res['log']))
    save_npz(os.path.join(out, 'f8_subscenes.npz'), **arrays)


def capture_f9(out):
    """get_H_diff_matrix (p2w:1945-1995) on a sparse scene: pairs with fewer than 10 common points,
    with 10..49 (SVD branch) and with 50 or more (ARPACK branch), both pixel normalisations."""
    sc = make_scene(260, 10, seed=41, vis_frac=0.6)
    X_pix = sc['X_pix'].copy()
    X_pix[40:, 7] = -1                  # a camera with few points
    X_pix[8:, 8] = -1                   # ... and one with fewer than 10
    X_pix[:130, 9] = -1
    vis = X_pix[:, :, 0] >= 0
    ii, jj = np.nonzero(vis)
    common = vis.T.astype(int) @ vis.astype(int)
    print('common points per pair: min', common.min(), 'max', common.max())
    arrays = dict(N=260, K=10, W=sc['W'], H=sc['H'], obs_i=ii.astype(np.int32), obs_j=jj.astype(np.int32),
                  obs_xy=X_pix[vis].astype(np.int32), common=common)
    arrays['H_diff_pix'] = p.get_H_diff_matrix(X_pix)
    arrays['H_diff_norm'] = p.get_H_diff_matrix(X_pix, normalize=True, W=sc['W'], H=sc['H'])
    save_npz(os.path.join(out, 'f9_h_diff.npz'), **arrays)


def capture_f10(out):
    """The three early exits of the loop (p2w:625-638) on the F3 scene: a run that ends on `stop_value`, two
    that end on `stop_diff` after learning-rate events (the iteration counter has gone down by then, so the
    window E[:iters - 2 * patience] is taken at a decremented counter), and one that reaches lr == 0.0 (an
    infinite factor: the first decrease divides the rate to zero; the parameters returned are the rolled-back
    ones)."""
    import warnings
    sc = make_scene(50, 12, seed=7, vis_frac=1.0)
    arrays = {k: sc[k] for k in ('X_pix', 'X_0', 'C_0', 'Theta_0')}
    arrays['W'], arrays['H'], arrays['phi_x_0'] = sc['W'], sc['H'], sc['phi_x_0']
    base = run_reference(sc, lr=1e-3, max_iters=200, patience=20)
    stop_value = 0.5 * (base['E'][61] + base['E'][62])
    runs = dict(stop_value=dict(lr=1e-3, max_iters=200, patience=20, stop_value=float(stop_value)),
                stop_diff_a=dict(lr=1e-3, max_iters=200, patience=3, stop_diff=10.0),
                stop_diff_b=dict(lr=1e-3, max_iters=200, patience=2, stop_diff=20.0),
                stop_diff_sgd=dict(lr=1e-7, max_iters=200, patience=2, stop_diff=50.0, optimizer='SGD'),
                lr_zero=dict(lr=1.0, max_iters=50, patience=20, factor=float('inf')))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for tag, kw in runs.items():
            r = run_reference(sc, **kw)
            print(tag, r['log'], r['E'].shape)
            for k, v in kw.items():
                arrays[f'{tag}__arg_{k}'] = v
            for k, v in r.items():
                arrays[f'{tag}__{k}'] = np.array(v) if k == 'log' else v
    arrays['runs'] = np.array(list(runs))
    save_npz(os.path.join(out, 'f10_early_stops.npz'), **arrays)


def _jacobian_recorder(keep, store):
    """frame_hook for run_reference: at the update steps listed in `keep`, copy the visible entries
    of the per-observation Jacobians the reference has just formed (p2w:654-657 D['X_model','X'],
    p2w:694-695 D['X_model','Theta'], p2w:700-701 D['X_model','phi_x']), the residuals (p2w:573) and
    the rotation matrices they were formed with (p2w:606-609, 711)."""
    def hook(step, loc):
        if step not in keep:
            return
        vis, D = loc['X_visible'], loc['D']
        store[step] = dict(J_X=D['X_model', 'X'][vis].copy(), J_Theta=D['X_model', 'Theta'][vis].copy(),
                           J_phi=D['X_model', 'phi_x'][vis].copy(), r=loc['X_diff'][vis].copy(),
                           R=np.array(loc['R']).copy(), Rx=np.array(loc['Rx']).copy(),
                           Ry=np.array(loc['Ry']).copy(), Rz=np.array(loc['Rz']).copy(),
                           f=np.array(loc['f'], dtype=np.float64).reshape(-1)[:1].copy())
    return hook


def capture_f11(out):
    """Per-observation Jacobians of the reference itself (the pin of the J^T J blocks, the Schur
    complement and the LM step, none of which the reference forms): the F1 scene (all groups free,
    Adam) at its first four iterates and the moscow_ride tracks at iterates 0 and 3."""
    arrays = {}
    sc = make_scene(60, 15, seed=3, vis_frac=0.6)
    vis = sc['X_pix'][:, :, 0] >= 0
    ii, jj = np.nonzero(vis)
    store = {}
    r = run_reference(sc, lr=1e-3, max_iters=5, frame_hook=_jacobian_recorder({0, 1, 2, 3}, store))
    arrays.update(f1_N=60, f1_K=15, f1_W=sc['W'], f1_H=sc['H'], f1_obs_i=ii.astype(np.int32),
                  f1_obs_j=jj.astype(np.int32), f1_obs_xy=sc['X_pix'][vis].astype(np.int32),
                  f1_keep=np.array(sorted(store)))
    for n in ('X', 'C', 'Theta', 'phi_x'):
        arrays['f1_step_' + n] = r['step_' + n][sorted(store)]
        arrays['f1_grad_' + n] = r['grad_' + n][sorted(store)]
    for k in store[0]:
        arrays['f1_' + k] = np.array([store[t][k] for t in sorted(store)])

    X_pix, W, H = read_tracks_pickle(os.path.join(REF_DIR, 'demo_examples/moscow_ride/input.txt'))
    X0, C0, Th0, phi0 = moscow_init(X_pix, W, H)
    sc = dict(X_pix=X_pix, W=W, H=H, X_0=X0, C_0=C0, Theta_0=Th0, phi_x_0=phi0)
    store = {}
    r = run_reference(sc, lr=1e-3, max_iters=5, patience=1000, frame_hook=_jacobian_recorder({0, 3}, store))
    arrays['f2_keep'] = np.array(sorted(store))       # observation list, sizes: f2_moscow_ride.npz
    for n in ('X', 'C', 'Theta', 'phi_x'):
        arrays['f2_step_' + n] = r['step_' + n][sorted(store)]
        arrays['f2_grad_' + n] = r['grad_' + n][sorted(store)]
    for k in store[0]:
        arrays['f2_' + k] = np.array([store[t][k] for t in sorted(store)])
    save_npz(os.path.join(out, 'f11_jacobians.npz'), **arrays)


def capture_f12(out):
    """Inputs the reference accepts without complaint and every other fixture avoids (p2w:571-573, 654-655 divide
    by the depth whatever its sign or size; visibility is X_pix[:,:,0] >= 0 alone, p2w:506):
      edge   observations of points BEHIND their cameras, pixels at the int16 limits, a negative y
      wide   the same with pixels beyond the int16 range
      tiny   one point at depth 2^-40 in cameras 0 and 1 (camera 0 at the origin with zero angles and the longest
             baseline exactly 1 make the initial gauge fix exact, so both observations have exactly that depth)
      zero   the same point at depth exactly 0: the reference's energy and gradients are inf / NaN
      blind  cameras and phi_x free, points fixed, one camera that sees nothing (no gauge, no connectivity test)
    Adam, teacher-forcing data of the first iterations of each."""
    import warnings
    rng = np.random.default_rng(77)
    N, K, W, H, phi = 28, 6, 1920, 1080, 1.2
    C = np.zeros((K, 3))
    C[1] = (1.0, 0.0, 0.0)
    C[2:] = rng.uniform(-0.4, 0.4, (K - 2, 3))
    C[2:, 2] -= 0.3
    assert np.linalg.norm(C[2:], axis=1).max() < 0.95
    Theta = np.zeros((K, 3))
    Theta[2:] = rng.normal(0, 0.05, (K - 2, 3))
    X = rng.uniform(-2, 2, (N, 3))
    X[:, 2] = rng.uniform(3, 6, N)
    X[1:4, 2] = -rng.uniform(3, 5, 3)                       # behind every camera
    X[0] = (0.5, 0.25, 1.0)                                 # depth set per case below
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = p.transform(X, C, Theta, phi, W, H, delete_back_points=False)
    base = np.floor(model).astype(np.int32)
    vis = rng.random((N, K)) < 0.75
    vis[:, 0] |= ~vis.any(axis=1)
    vis[1:4, 2:5] = True                                    # the points behind the cameras are observed
    vis[0, :] = [False, False, True, True, False, True]
    base[~vis] = -1
    base[vis & (base[:, :, 0] < 0), 0] = 3                  # keep every chosen pair visible (x >= 0)
    edge = base.copy()
    edge[5, 2], edge[6, 3], edge[7, 4], edge[8, 2] = (32767, -32768), (0, 32767), (32767, 0), (11, -7)
    for ij in ((5, 2), (6, 3), (7, 4), (8, 2)):
        vis[ij] = True
    wide = edge.copy()
    wide[5, 2], wide[6, 3], wide[7, 4] = (40000, -40000), (70000, 12), (32768, -32769)
    tiny = edge.copy()
    tiny[0, 0], tiny[0, 1] = (700, 300), (1200, 310)
    blind = edge.copy()
    blind[:, 5] = -1
    rng2 = np.random.default_rng(78)
    X0 = X + rng2.normal(0, 0.05, X.shape)
    C0 = C.copy()
    C0[2:] += rng2.normal(0, 0.02, (K - 2, 3))
    Th0 = Theta.copy()
    Th0[2:] += rng2.normal(0, 0.01, (K - 2, 3))
    X0_tiny, X0_zero = X0.copy(), X0.copy()
    X0_tiny[0] = (0.5, 0.25, 2.0 ** -40)
    X0_zero[0] = (0.5, 0.25, 0.0)
    cases = dict(edge=(edge, X0, (True, True, True), 4), wide=(wide, X0, (True, True, True), 4),
                 tiny=(tiny, X0_tiny, (True, True, True), 4), zero=(tiny, X0_zero, (True, True, True), 2),
                 blind=(blind, X0, (False, True, True), 4))
    arrays = dict(N=N, K=K, W=W, H=H, C_0=C0, Theta_0=Th0, phi_x_0=phi * 1.02, cases=np.array(list(cases)))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for tag, (X_pix, X_init, masks, iters) in cases.items():
            sc = dict(X_pix=X_pix, W=W, H=H, X_0=X_init, C_0=C0, Theta_0=Th0, phi_x_0=phi * 1.02)
            r = run_reference(sc, lr=1e-3, max_iters=iters, masks=masks)
            arrays[f'{tag}__X_pix'], arrays[f'{tag}__X_0'], arrays[f'{tag}__masks'] = X_pix, X_init, np.array(masks)
            for k, v in r.items():
                arrays[f'{tag}__{k}'] = np.array(v) if k == 'log' else v
            print(tag, 'E', r['E'], 'steps', r['grad_C'].shape[0],
                  'non-finite gradient rows (C)', (~np.isfinite(r['grad_C'][0])).any(axis=1).sum())
    save_npz(os.path.join(out, 'f12_edge_cases.npz'), **arrays)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden'))
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    todo = dict(f1=capture_f1, f2=capture_f2, f3=capture_f3, f4=capture_f4, f5=capture_f5,
                f6=capture_f6, f7=capture_f7, f8=capture_f8, f9=capture_f9, f10=capture_f10,
                f11=capture_f11, f12=capture_f12)
    for name, fn in todo.items():
        if a.only and name not in a.only.split(','):
            continue
        fn(a.out)


if __name__ == '__main__':
    main()
