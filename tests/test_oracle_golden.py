"""Pins the CPU oracle (oracle/ba_oracle.py) against vectors captured from the reference
itself (oracle/capture_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, f11_case, f12_check, logs_match, normal_equations_from_jacobians, rel_err
from oracle import ba_oracle as orc

TOL = 1e-9       # per-iteration energy / gradient contract (relative, max-norm)


def test_f5_rotations_gauge_adam(golden):
    g = golden('f5_gauge_funcs.npz')
    Th = g['Theta']
    assert np.array_equal(orc.rx(Th[:, 0]), g['Rx'])
    assert np.array_equal(orc.ry(Th[:, 1]), g['Ry'])
    assert np.array_equal(orc.rz(Th[:, 2]), g['Rz'])
    assert np.array_equal(orc.ryxz(Th), g['R'])
    assert np.array_equal(orc.get_rotations(g['R']), g['Theta_back'])
    Xa, Ca, Tha, Ra = orc.align_scene(g['X'], g['C'], Theta=Th, ret_R=True)
    for a, b in ((Xa, g['align_X']), (Ca, g['align_C']), (Tha, g['align_Theta']), (Ra, g['align_R'])):
        assert np.array_equal(a, b)
    Xn, Cn, j1 = orc.normalize_scene(Xa, Ca, ret_j=True)
    assert j1 == int(g['norm_j1'])
    assert np.array_equal(Xn, g['norm_X']) and np.array_equal(Cn, g['norm_C'])
    xa, st = orc.gd_adam(g['adam_x'], g['adam_g'], 1e-2, g['adam_s'], g['adam_r'], np.array([3]))
    assert np.array_equal(xa, g['adam_x_new'])
    assert np.array_equal(st['s'], g['adam_s_new']) and np.array_equal(st['r'], g['adam_r_new'])
    assert int(st['t'][0]) == int(g['adam_t_new'][0])
    assert np.array_equal(orc.sgd(g['adam_x'], g['adam_g'], 1e-2)[0], g['sgd_x_new'])


def test_f5_transform_and_error(golden):
    g = golden('f5_gauge_funcs.npz')
    W, H, phi = int(g['tr_W']), int(g['tr_H']), float(g['tr_phi'])
    args = (g['tr_X'], g['tr_C'], g['Theta'], phi, W, H)
    for kw, key in ((dict(), 'tr_default'), (dict(delete_boundary_points=True), 'tr_boundary'),
                    (dict(delete_back_points=False), 'tr_keep_back')):
        got, want = orc.transform_dense(*args, **kw), g[key]
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert np.allclose(got, want, rtol=1e-12, atol=1e-9, equal_nan=True)
    assert orc.error(g['err_X_pix'], g['tr_default']) == pytest.approx(float(g['err_value']), rel=1e-13)
    shifted = orc.transform_dense(g['tr_X'] + 0.01, g['tr_C'], g['Theta'], phi, W, H)
    assert orc.error(g['err_X_pix'], shifted) == pytest.approx(float(g['err_value_shifted']), rel=1e-12)


def _f7_cases(g):
    for tag in ('true', 'noisy'):
        for norm in (False, True):
            yield tag, norm, 'norm' if norm else 'pix'


def test_f7_triangulation_and_resection(golden):
    """Linear initialisation steps (p2w:1655-1808) restated on the visible pairs vs the reference's
    own outputs: same NaN pattern (points seen < 2 times, cameras with < 6 points), values to
    rounding for the triangulation and the SVD resection; the Gram-matrix resection (the
    reference's path for large inputs) within what squaring the condition number allows."""
    g = golden('f7_linear_init.npz')
    N, K, W, H, phi = int(g['N']), int(g['K']), int(g['W']), int(g['H']), float(g['phi_x'])
    i, j, uv = g['obs_i'], g['obs_j'], g['obs_xy'] + 0.5
    for tag, norm, n in _f7_cases(g):
        X, C, Th = g[f'X_{tag}'], g[f'C_{tag}'], g[f'Theta_{tag}']
        got = orc.triangulate_obs(i, j, uv, N, C, Th, phi, W, H, norm)
        want = g[f'tri_{tag}_{n}']
        assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(want[:4]).all()
        assert np.nanmax(np.abs(got - want)) < 1e-12
        for gram, tol in ((False, 1e-10), (True, 1e-6 if not norm else 1e-10)):
            Cg, Tg = orc.resect_obs(i, j, uv, K, X, phi, W, H, norm, use_gram=gram)
            assert np.array_equal(np.isnan(Cg), np.isnan(g[f'res_C_{tag}_{n}'])) and np.isnan(Cg[5]).all()
            assert np.nanmax(np.abs(Cg - g[f'res_C_{tag}_{n}'])) < tol
            assert np.nanmax(np.abs(Tg - g[f'res_Theta_{tag}_{n}'])) < tol
    Cg, Tg = orc.resect_obs(i, j, uv, K, g['X_with_nan'], phi, W, H, True, use_gram=False)
    assert np.nanmax(np.abs(Cg - g['res_C_nan_norm'])) < 1e-10 and np.nanmax(np.abs(Tg - g['res_Theta_nan_norm'])) < 1e-10


def test_visibility_indexing_bit_exact(golden):
    g = golden('f2_moscow_ride.npz')
    N, K = int(g['N']), int(g['K'])
    X_pix = np.full((N, K, 2), -1, dtype=np.int32)
    X_pix[g['obs_i'], g['obs_j']] = g['obs_xy']
    i, j, uv = orc.visibility_coo(X_pix)
    assert np.array_equal(i, g['obs_i']) and np.array_equal(j, g['obs_j'])
    assert np.array_equal(uv, g['obs_xy'] + 0.5)
    assert i.size == 7933
    assert orc.count_components(i, j, N, K) == 1


def test_f4_lr_manager_scripted():
    with open(os.path.join(GOLDEN, 'f4_lr_manager.json')) as fh:
        cases = json.load(fh)
    for case in cases:
        E, patience, factor = np.array(case['E']), case['patience'], case['factor']
        n = E.size
        params = {'a': {'main': np.zeros(3)}, 'b': {'main': np.zeros(1)}}
        orc.form_params_dict(params, 'Adam', True, 4 * n + 8)
        calls = [0]

        def recompute():
            calls[0] += 1
            return ('x',), {}

        lr, iters = 1.0, 0
        E_arr = np.zeros(4 * n + 8)
        E_min = E_min_temp = np.inf
        pt, st, began, inc = 0, patience, False, False
        log = []
        for step in range(n):
            E_arr[iters] = E[step]
            for val in params.values():
                val['main'] += 1.0
                val['gd']['s'] += 0.5
                val['gd']['t'] += 1
            tup, lst = orc.lr_manager(lr, iters, E_arr, E_min, E_min_temp, pt, st, began, inc,
                                      patience, factor, params, True, recompute, log=log)
            lr, iters, E_min, E_min_temp, pt, st, began, inc = tup
            want = case['trace'][step]
            got = dict(lr=float(lr), iters=int(iters), E_min=float(E_min),
                       E_min_temp=float(E_min_temp), patience_timer=int(pt), success_timer=int(st),
                       began=bool(began), increased=bool(inc), recomputed=len(lst),
                       main_a=params['a']['main'].tolist(), release_a=params['a']['release'].tolist(),
                       s_a=params['a']['gd']['s'].tolist(), t_a=int(params['a']['gd']['t'][0]))
            assert got == want, (step, got, want)
            iters += 1
        assert log == case['log']
        assert calls[0] == case['recompute_calls']


def _coo(X_pix):
    return orc.visibility_coo(X_pix)


@pytest.mark.parametrize('opt', ['Adam', 'SGD'])
def test_f1_all_mask_combinations(golden, opt):
    g = golden('f1_tiny_masks.npz')
    X_pix = g['X_pix']
    N, K = X_pix.shape[:2]
    i, j, uv = _coo(X_pix)
    W, H = int(g['W']), int(g['H'])
    lr = 1e-3 if opt == 'Adam' else 1e-7
    for m in g['combos']:
        m = tuple(bool(t) for t in m)
        tag = f"{opt}_{''.join('1' if t else '0' for t in m)}"
        glog, log = [], []
        out = orc.fit_scene(i, j, uv, N, K, W, H, lr, 5, g['X_0'], g['C_0'], g['Theta_0'],
                            float(g['phi_x_0']), X_mask=m[0], C_Theta_mask=m[1], phi_x_mask=m[2],
                            optimizer=opt, log=log, grad_log=glog)
        names = (['X'] if m[0] else []) + (['C', 'Theta'] if m[1] else []) + (['phi_x'] if m[2] else [])
        assert len(out) == len(names) + 1
        assert rel_err(out[-1], g[f'{tag}__E']) < TOL
        for n, v in zip(names, out):
            assert np.abs(np.asarray(v) - g[f'{tag}__final_{n}']).max() < 1e-9, (tag, n)
        per_step = len(names)
        for s in range(len(glog) // per_step):
            for k, n in enumerate(names):
                name, x, grad = glog[s * per_step + k]
                assert name == n
                assert np.abs(x - g[f'{tag}__step_{n}'][s]).max() < 1e-9
                assert rel_err(grad, g[f'{tag}__grad_{n}'][s]) < TOL, (tag, n, s)
        assert log == [ln for ln in g[f'{tag}__log'].tolist() if 'Reached' in ln]


def test_f2_moscow_teacher_forced(golden):
    g = golden('f2_moscow_ride.npz')
    W, H = int(g['W']), int(g['H'])
    i, j = g['obs_i'], g['obs_j']
    uv = g['obs_xy'] + 0.5
    steps = g['tf__step_X'].shape[0]
    assert steps == 50
    worst = 0.0
    for s in range(steps):
        E, gX, gC, gT, gp = orc.energy_and_gradients(
            g['tf__step_X'][s], g['tf__step_C'][s], g['tf__step_Theta'][s], g['tf__step_phi_x'][s],
            W, H, i, j, uv)
        errs = (abs(E - g['tf__E'][s]) / g['tf__E'][s], rel_err(gX, g['tf__grad_X'][s]),
                rel_err(gC, g['tf__grad_C'][s]), rel_err(gT, g['tf__grad_Theta'][s]),
                rel_err(gp, g['tf__grad_phi_x'][s]))
        worst = max(worst, *errs)
    assert worst < TOL, worst


def test_f2_moscow_free_run_300(golden):
    g = golden('f2_moscow_ride.npz')
    W, H = int(g['W']), int(g['H'])
    log = []
    X, C, Th, phi, E = orc.fit_scene(g['obs_i'], g['obs_j'], g['obs_xy'] + 0.5, int(g['N']), int(g['K']),
                                     W, H, 1e-3, 300, g['free__X_0'], g['C_0'], g['Theta_0'],
                                     float(g['phi_x_0']), patience=1000, log=log)
    assert E.shape == g['free__E'].shape
    assert rel_err(E, g['free__E']) < 1e-8
    assert np.abs(X - g['free__final_X']).max() < 1e-6
    assert np.abs(C - g['free__final_C']).max() < 1e-6
    assert np.abs(Th - g['free__final_Theta']).max() < 1e-6
    assert abs(phi - float(g['free__final_phi_x'])) < 1e-6
    assert log == ['299 : Reached maximum iterations']


def test_f3_lr_events(golden):
    g = golden('f3_lr_events.npz')
    X_pix = g['X_pix']
    N, K = X_pix.shape[:2]
    i, j, uv = _coo(X_pix)
    log, glog = [], []
    out = orc.fit_scene(i, j, uv, N, K, int(g['W']), int(g['H']), 1e-3, 200, g['X_0'], g['C_0'],
                        g['Theta_0'], float(g['phi_x_0']), patience=20, log=log, grad_log=glog)
    want = [ln for ln in g['log'].tolist() if ('LR' in ln or 'Reached' in ln)]
    assert log == want
    assert len(glog) // 4 == g['step_X'].shape[0]       # same number of executed updates
    E = out[-1]
    assert E.shape == g['E'].shape
    assert rel_err(E[:100], g['E'][:100]) < 1e-8
    # teacher-forced gradients at the reference's own iterates, rollback steps included
    for s in range(0, g['step_X'].shape[0], 7):
        _, gX, gC, gT, gp = orc.energy_and_gradients(g['step_X'][s], g['step_C'][s], g['step_Theta'][s],
                                                     g['step_phi_x'][s], int(g['W']), int(g['H']), i, j, uv)
        assert rel_err(gX, g['grad_X'][s]) < TOL and rel_err(gC, g['grad_C'][s]) < TOL
        assert rel_err(gT, g['grad_Theta'][s]) < TOL and rel_err(gp, g['grad_phi_x'][s]) < TOL


def _f10_kwargs(g, tag):
    kw = {}
    for key in g.files:
        if key.startswith(tag + '__arg_'):
            v = g[key]
            kw[key[len(tag) + 6:]] = str(v) if v.dtype.kind == 'U' else (int(v) if v.dtype.kind == 'i' else float(v))
    return kw


def test_f10_early_stops(golden):
    """The three early exits of the loop (p2w:625-638): stop_value, stop_diff after learning-rate events
    (iteration counter already decremented), lr == 0.0 - printed line, length of E, returned parameters."""
    g = golden('f10_early_stops.npz')
    X_pix = g['X_pix']
    N, K = X_pix.shape[:2]
    i, j, uv = _coo(X_pix)
    seen = set()
    for tag in g['runs'].tolist():
        kw = _f10_kwargs(g, tag)
        lr, max_iters = kw.pop('lr'), int(kw.pop('max_iters'))
        log = []
        out = orc.fit_scene(i, j, uv, N, K, int(g['W']), int(g['H']), lr, max_iters, g['X_0'], g['C_0'], g['Theta_0'],
                            float(g['phi_x_0']), log=log, **kw)
        want = [ln for ln in g[f'{tag}__log'].tolist() if ('LR' in ln or 'Reached' in ln)]
        assert logs_match(log, want), (tag, log, want)
        seen.add(want[-1].split(' : ')[1].split(':')[0].split(' <')[0])
        assert out[-1].shape == g[f'{tag}__E'].shape, tag
        assert rel_err(out[-1], g[f'{tag}__E']) < 1e-8, tag
        for n, v in zip(('X', 'C', 'Theta', 'phi_x'), out):
            assert np.abs(np.asarray(v) - g[f'{tag}__final_{n}']).max() < 1e-8, (tag, n)
    assert seen == {'Reached stopping value', 'Reached stopping difference', 'Reached lr = 0.0'}


def test_f6_medium_scene(golden):
    g = golden('f6_medium.npz')
    i, j = g['obs_i'], g['obs_j']
    uv = g['obs_xy'] + 0.5
    W, H = int(g['W']), int(g['H'])
    for n, s in enumerate(g['keep']):
        E, gX, gC, gT, gp = orc.energy_and_gradients(g['step_X'][n], g['step_C'][n], g['step_Theta'][n],
                                                     g['step_phi_x'][n], W, H, i, j, uv)
        assert rel_err(gX, g['grad_X'][n]) < TOL and rel_err(gC, g['grad_C'][n]) < TOL
        assert rel_err(gT, g['grad_Theta'][n]) < TOL and rel_err(gp, g['grad_phi_x'][n]) < TOL
    log = []
    out = orc.fit_scene(i, j, uv, int(g['N']), int(g['K']), W, H, 1e-3, 500, g['X_0'], g['C_0'],
                        g['Theta_0'], float(g['phi_x_0']), patience=100, log=log)
    want = [ln for ln in g['log'].tolist() if ('LR' in ln or 'Reached' in ln)]
    assert log == want
    assert rel_err(out[-1][:100], g['E'][:100]) < 1e-8
    assert np.abs(out[0] - g['final_X']).max() < 1e-3      # 500 free iterations with 8 LR events


def test_c_oracle_matches_golden_and_numpy_oracle(golden):
    """The C restatement (CPU baseline of bench.py) against the reference's vectors."""
    from oracle import c_oracle
    g = golden('f2_moscow_ride.npz')
    W, H = int(g['W']), int(g['H'])
    i, j, uv = g['obs_i'], g['obs_j'], g['obs_xy'] + 0.5
    for s in (0, 7, 49):
        E, gX, gC, gT, gp = c_oracle.energy_and_gradients(g['tf__step_X'][s], g['tf__step_C'][s],
                                                          g['tf__step_Theta'][s], g['tf__step_phi_x'][s][0], W, H, i, j, uv)
        assert abs(E - g['tf__E'][s]) / g['tf__E'][s] < TOL
        assert rel_err(gX, g['tf__grad_X'][s]) < TOL and rel_err(gC, g['tf__grad_C'][s]) < TOL
        assert rel_err(gT, g['tf__grad_Theta'][s]) < TOL and rel_err(gp, g['tf__grad_phi_x'][s]) < TOL
    # full iterations: same trajectory as the reference while no LR event fires
    it = c_oracle.Iterator(i, j, uv, g['free__X_0'], g['C_0'], g['Theta_0'], float(g['phi_x_0']), W, H, 1e-3)
    E = np.array([it.step() for _ in range(60)])
    assert rel_err(E, g['free__E'][:60]) < 1e-9


# ---- F8: joint alignment of subscenes (p2w:729-949) ----------------------------------------------
F8_RUNS = dict(adam=dict(lr=1e-2, max_iters=40, optimizer='Adam'),
               sgd=dict(lr=1e-2, max_iters=40, optimizer='SGD'),
               start=dict(lr=1e-3, max_iters=300, optimizer='Adam', start=True),
               events=dict(lr=2e-3, max_iters=56, optimizer='Adam', start=True, patience=12))


def f8_kwargs(g, tag):
    kw = dict(F8_RUNS[tag])
    if kw.pop('start', False):
        kw.update(S_0=g['S_0'], T_0=g['T_0'], Psi_0=g['Psi_0'])
    return kw


def test_f8_subscene_gradients_teacher_forced(golden):
    """Energy and the three gradients at the reference's own iterates."""
    g = golden('f8_subscenes.npz')
    X = g['X']
    N, P = X.shape[:2]
    i, j, x = orc.subscene_entries(X)
    for tag in ('adam', 'sgd', 'start'):
        S_, T_, Psi_ = g[f'{tag}_step_S'], g[f'{tag}_step_T'], g[f'{tag}_step_Psi']
        for t in range(0, S_.shape[0], 7):
            E, gS, gT, gPsi = orc.subscenes_energy_and_gradients(i, j, x, N, P, S_[t], T_[t], Psi_[t])
            assert abs(E - g[f'{tag}_E'][t]) <= 1e-12 * g[f'{tag}_E'][t]
            assert rel_err(gS, g[f'{tag}_grad_S'][t]) < 1e-11
            assert rel_err(gT, g[f'{tag}_grad_T'][t]) < 1e-11
            assert rel_err(gPsi, g[f'{tag}_grad_Psi'][t]) < 1e-11


@pytest.mark.parametrize('tag', list(F8_RUNS))
def test_f8_subscene_runs(golden, tag):
    """Free runs: energies, every update's parameters and gradients, the event log and the results."""
    g = golden('f8_subscenes.npz')
    kw = f8_kwargs(g, tag)
    log, trace = [], []
    out = orc.fit_subscenes(g['X'], kw.pop('lr'), kw.pop('max_iters'), log=log, trace=trace, **kw)
    E_ref = g[f'{tag}_E']
    assert out[4].shape == E_ref.shape
    assert rel_err(out[4], E_ref) < 1e-9
    assert len(trace) == g[f'{tag}_step_S'].shape[0]
    for name, k in (('S', 1), ('T', 2), ('Psi', 3)):
        got = np.array([t[k] for t in trace]).reshape(g[f'{tag}_step_{name}'].shape)
        assert np.abs(got - g[f'{tag}_step_{name}']).max() < 1e-9
        assert np.abs(out[k].reshape(g[f'{tag}_final_{name}'].shape) - g[f'{tag}_final_{name}']).max() < 1e-9
    np.testing.assert_allclose(out[0], g[f'{tag}_X_new'], rtol=0, atol=1e-9, equal_nan=True)
    events = [ln for ln in g[f'{tag}_log'].tolist() if 'LR to' in ln]
    assert logs_match(log, events)


# ---- F9: homography misfit of the camera pairs (p2w:1945-1995) -------------------------------------
def test_f9_h_diff_matrix(golden):
    g = golden('f9_h_diff.npz')
    uv = g['obs_xy'] + 0.5
    W, H = float(g['W']), float(g['H'])
    for tag, norm in (('pix', False), ('norm', True)):
        ref = g['H_diff_' + tag]
        got = orc.h_diff_matrix_obs(g['obs_i'], g['obs_j'], uv, int(g['K']), norm, W, H)
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        ok = ~np.isnan(ref)
        assert (np.abs(got - ref)[ok] <= 1e-9 * ref[ok]).all()


@pytest.mark.parametrize('which,t', [('f1', 0), ('f1', 1), ('f1', 2), ('f1', 3), ('f2', 0), ('f2', 1)])
def test_f11_jacobians_of_the_reference(golden, which, t):
    """The restatement's per-observation Jacobian blocks against the arrays the reference itself formed inside
    fit_scene (p2w:654-657, 694-695, 700-701), with the rotation matrices the reference used at that iterate; and
    the gradients the reference handed to its update rule are J^T r / |V| of those very blocks."""
    c = f11_case(golden, which, t)
    r, J_X, J_Theta, J_phi = orc.jacobian_blocks(c['step_X'], c['step_C'], c['R'], c['Rx'], c['Ry'], c['Rz'],
                                                 float(c['f'][0]), float(c['step_phi_x'][0]), c['W'], c['H'],
                                                 c['i'], c['j'], c['uv'])
    assert rel_err(r, c['r']) < 1e-12
    assert rel_err(J_X, c['J_X']) < 1e-12
    assert rel_err(J_Theta, c['J_Theta']) < 1e-12
    assert rel_err(J_phi, c['J_phi']) < 1e-12
    # dense_jacobian (angles -> R by sincos, what the second-order tests use) against the same arrays
    rr, J = orc.dense_jacobian(c['step_X'], c['step_C'], c['step_Theta'], c['step_phi_x'], c['W'], c['H'],
                               c['i'], c['j'], c['uv'])
    N, K, nV = c['N'], c['K'], c['i'].size
    o = np.arange(nV)
    for a in range(2):
        for b in range(3):
            assert rel_err(J[2 * o + a, 3 * c['i'] + b], c['J_X'][:, a, b]) < 1e-10
            assert rel_err(J[2 * o + a, 3 * N + 3 * K + 3 * c['j'] + b], c['J_Theta'][:, a, b]) < 1e-10
        assert rel_err(J[2 * o + a, -1], c['J_phi'][:, a]) < 1e-10
    ne = normal_equations_from_jacobians(c)
    g = ne['g'] / nV
    assert rel_err(g[:3 * N].reshape(N, 3), c['grad_X']) < 1e-12
    gcam = g[3 * N:3 * N + 6 * K].reshape(K, 6)
    assert rel_err(gcam[:, :3], c['grad_C']) < 1e-12 and rel_err(gcam[:, 3:], c['grad_Theta']) < 1e-12
    assert abs(g[-1] - c['grad_phi_x'][0]) < 1e-12 * abs(c['grad_phi_x'][0])


def test_f12_edge_cases_teacher_forced(golden):
    """The restatement at the reference's iterates on the inputs of golden F12 (see conftest.f12_check)."""
    import warnings
    g = golden('f12_edge_cases.npz')
    W, H = float(g['W']), float(g['H'])
    for tag in g['cases'].tolist():
        i, j, uv = orc.visibility_coo(g[f'{tag}__X_pix'])
        masks = g[f'{tag}__masks']
        steps = g[f'{tag}__grad_C'].shape[0]
        assert steps == (1 if tag == 'zero' else 3)
        for s_ in range(steps):
            X = g[f'{tag}__step_X'][s_] if masks[0] else g[f'{tag}__X_0']
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                E, gX, gC, gT, gp = orc.energy_and_gradients(X, g[f'{tag}__step_C'][s_], g[f'{tag}__step_Theta'][s_],
                                                             g[f'{tag}__step_phi_x'][s_], W, H, i, j, uv)
            f12_check(g, tag, s_, E, gX, gC, gT, gp)
    # what the cases are about really is in them
    g_t = golden('f12_edge_cases.npz')
    assert g_t['tiny__step_X'][0][0][2] == 2.0 ** -40 and g_t['tiny__E'][0] > 1e27
    assert np.isinf(g_t['zero__E'][0]) and (g_t['blind__X_pix'][:, 5, 0] < 0).all()
    assert g_t['wide__X_pix'].max() == 70000 and g_t['edge__X_pix'][:, :, 1].min() == -32768
