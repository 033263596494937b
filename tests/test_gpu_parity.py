"""Parity of the HIP path (through the C ABI) with the CPU oracle and with the golden vectors
captured from the reference.  Needs an MI355X: run with `-m gpu`.

Tolerances (the contract): energy and gradients within 1e-9 relative per evaluation at the
reference's own iterates (teacher-forced), final parameters within 1e-6 after a fixed free run,
visibility indexing bit-exact."""
import io
import os
import contextlib

import numpy as np
import pytest

from conftest import f12_check, logs_match, rel_err, row_err
from oracle import ba_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-9
DEV = 'cuda:0'


def _dense_from_coo(g):
    N, K = int(g['N']), int(g['K'])
    X_pix = np.full((N, K, 2), -1, dtype=np.int32)
    X_pix[g['obs_i'], g['obs_j']] = g['obs_xy']
    return X_pix


# ---- H0: visibility extraction ---------------------------------------------------------------
@pytest.mark.parametrize('shape,frac,seed', [((93, 327), 0.26, 0), ((1, 1), 1.0, 1), ((7, 64), 0.5, 2),
                                            ((130, 65), 0.0, 3), ((257, 129), 1.0, 4), ((9000, 37), 0.3, 5)])
def test_visibility_random_bit_exact(ba, shape, frac, seed):
    rng = np.random.default_rng(seed)
    N, K = shape
    X_pix = rng.integers(0, 4000, (N, K, 2)).astype(np.int32)
    X_pix[rng.random((N, K)) >= frac] = -1
    oi, oj, uv, row_ptr = ba.engine.visibility_coo(X_pix, DEV)
    i, j, uv_ref = orc.visibility_coo(X_pix)
    assert np.array_equal(oi.cpu().numpy(), i) and np.array_equal(oj.cpu().numpy(), j)
    assert np.array_equal(uv.cpu().numpy().reshape(-1, 2), uv_ref.reshape(-1, 2))
    counts = (X_pix[:, :, 0] >= 0).sum(1)
    assert np.array_equal(row_ptr.cpu().numpy(), np.concatenate([[0], np.cumsum(counts)]))


def test_visibility_moscow_bit_exact(ba, golden):
    g = golden('f2_moscow_ride.npz')
    oi, oj, uv, _ = ba.engine.visibility_coo(_dense_from_coo(g), DEV)
    assert np.array_equal(oi.cpu().numpy(), g['obs_i']) and np.array_equal(oj.cpu().numpy(), g['obs_j'])
    assert np.array_equal(uv.cpu().numpy(), g['obs_xy'] + 0.5)


def test_visibility_only_x_decides(ba):
    # the rule is X_pix[:,:,0] >= 0 alone (p2w:506): a negative y with x >= 0 is visible
    X_pix = np.array([[[5, -1], [-1, 7], [0, 0]]], dtype=np.int32)
    oi, oj, uv, _ = ba.engine.visibility_coo(X_pix, DEV)
    assert oj.cpu().numpy().tolist() == [0, 2]
    assert uv.cpu().numpy().tolist() == [[5.5, -0.5], [0.5, 0.5]]


# ---- H1: the visibility graph's components, counted on the device (p2w:519-525) ------------------------
def test_visibility_components_match_union_find(ba):
    """Random bipartite graphs (isolated points and cameras, many small components), a long chain (one camera pair
    per point: the worst case for label propagation) and a video orbit: the device count equals the union-find
    count of the restatement, which is the count scipy gives the reference on the dense adjacency."""
    rng = np.random.default_rng(0)
    cases = []
    for N, K, n in ((30, 8, 40), (50, 20, 45), (10, 3, 30), (2000, 700, 1500), (1, 1, 1), (5, 4, 0)):
        cases.append((np.sort(rng.integers(0, N, n)).astype(np.int32), rng.integers(0, K, n).astype(np.int32), N, K))
    K = 3000                                            # chain: point p sees cameras p and p + 1
    i = np.repeat(np.arange(K - 1), 2).astype(np.int32)
    j = (i + np.tile([0, 1], K - 1)).astype(np.int32)
    cases.append((i, j, K - 1, K))
    cases.append((i[:-2], j[:-2], K - 1, K))            # ... with the last point unobserved and the last camera blind
    sc = ba.scenes.orbit_scene(N=20000, K=300, track_len=10, seed=1)
    cases.append((sc['i'], sc['j'], sc['N_local'], 300))
    for i, j, N, K in cases:
        n_cam, n_pts, labels = ba.engine.visibility_components(i, j, N, K, DEV, want_labels=True)
        assert n_cam + n_pts == orc.count_components(i, j, N, K), (N, K, i.size)
        assert n_pts == int((np.bincount(i, minlength=N) == 0).sum())
        lab = labels.cpu().numpy()
        assert (lab <= np.arange(K)).all() and (lab[lab] == lab).all() and np.unique(lab).size == n_cam
    assert ba.pix2world.visibility_graph_components(cases[-1][0], cases[-1][1], cases[-1][2], 300, DEV) == 1


def test_point_order_is_the_stable_sort_by_first_camera(ba):
    """The engine's internal point order, built on the device (key, stable radix sort, rewrite of the list):
    ascending first observing camera, ties and every point's observations in the caller's order; unobserved
    points last; nothing is permuted when the list already has the property; an unsorted list is refused."""
    import torch
    rng = np.random.default_rng(3)
    for N, K, deg in ((5000, 40, 6), (3000, 300, 3), (70000, 70000, 2), (100, 2, 2)):
        counts = rng.integers(0, deg + 1, N)
        i = np.repeat(np.arange(N), counts).astype(np.int32)
        j = np.concatenate([np.sort(rng.choice(K, c, replace=False)) for c in counts] + [np.zeros(0, int)]).astype(np.int32)
        uv = rng.normal(0, 1, (i.size, 2))
        sh = ba.engine.Shard(i, j, uv, N, K, 640, 480, DEV)
        first = np.full(N, K)
        np.minimum.at(first, i, j)
        want = np.argsort(first, kind='stable')
        if (np.diff(first) >= 0).all():
            assert sh.perm is None
            continue
        assert np.array_equal(sh.perm.cpu().numpy(), want)
        # the observation list in that order, through a round trip of per-point data
        marks = torch.arange(N, dtype=torch.float64, device=DEV)
        assert torch.equal(sh.points_to_caller(sh.points_to_engine(marks.view(-1, 1).repeat(1, 3))), marks.view(-1, 1).repeat(1, 3))
        sh.close()
    sc = ba.scenes.orbit_scene(N=3000, K=60, track_len=8, seed=2)      # tracker output: already ordered
    sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], 60, sc['W'], sc['H'], DEV)
    first = np.full(sc['N_local'], 60)
    np.minimum.at(first, sc['i'], sc['j'])
    assert (sh.perm is None) == bool((np.diff(first) >= 0).all())
    sh.close()
    with pytest.raises(ba.engine._lib.BAError, match='sorted by point'):
        ba.engine.Shard(np.array([1, 0, 0], dtype=np.int32), np.array([0, 0, 1], dtype=np.int32), np.zeros((3, 2)), 2, 2, 640, 480, DEV)


# ---- H3/H4/H7/H8/H9: fused observation pass --------------------------------------------------
def _eval_all(ba, sh, nv, **kw):
    sh.eval(**kw)
    return (sh.energy(), sh.points_to_caller(sh.gX).cpu().numpy() / nv, sh.gC.cpu().numpy() / nv,
            sh.gTheta.cpu().numpy() / nv,
            sh.gphi.cpu().numpy() / nv)


@pytest.mark.parametrize('variant', [dict(), dict(point_major=True), dict(untiled=True), dict(cam_atomics=True)],
                         ids=['camera-runs', 'point-major-tiles', 'wave-chunks', 'global-atomics'])
def test_eval_teacher_forced_moscow(ba, golden, variant):
    g = golden('f2_moscow_ride.npz')
    W, H = int(g['W']), int(g['H'])
    nv = g['obs_i'].size
    sh = ba.engine.Shard(g['obs_i'], g['obs_j'], g['obs_xy'] + 0.5, int(g['N']), int(g['K']), W, H, DEV)
    worst = 0.0
    for s in range(g['tf__step_X'].shape[0]):
        sh.set_params(g['tf__step_X'][s], g['tf__step_C'][s], g['tf__step_Theta'][s], g['tf__step_phi_x'][s][0])
        E, gX, gC, gT, gp = _eval_all(ba, sh, nv, **variant)
        worst = max(worst, abs(E - g['tf__E'][s]) / g['tf__E'][s], rel_err(gX, g['tf__grad_X'][s]),
                    rel_err(gC, g['tf__grad_C'][s]), rel_err(gT, g['tf__grad_Theta'][s]),
                    rel_err(gp, g['tf__grad_phi_x'][s]))
        # every point and every camera on its own scale (camera 0, the gauge camera, included)
        worst = max(worst, row_err(gX, g['tf__grad_X'][s]), row_err(gC, g['tf__grad_C'][s]),
                    row_err(gT, g['tf__grad_Theta'][s]))
    sh.close()
    assert worst < TOL, worst


@pytest.mark.parametrize('variant', [dict(), dict(point_major=True), dict(untiled=True), dict(cam_atomics=True)],
                         ids=['camera-runs', 'point-major-tiles', 'wave-chunks', 'global-atomics'])
def test_eval_teacher_forced_edge_cases(ba, golden, variant):
    """Golden F12 through every kernel form: observations behind the camera, pixels at the int16 limits (8-byte
    records) and beyond them (f64 records), a depth of 2^-40, a depth of exactly 0 (same non-finite rows as the
    reference), a camera without observations; at the reference's own iterates."""
    g = golden('f12_edge_cases.npz')
    N, K, W, H = int(g['N']), int(g['K']), int(g['W']), int(g['H'])
    for tag in g['cases'].tolist():
        oi, oj, uv, _ = ba.engine.visibility_coo(g[f'{tag}__X_pix'], DEV)
        nv = int(oi.numel())
        sh = ba.engine.Shard(oi, oj, uv, N, K, W, H, DEV)
        m = [bool(t) for t in g[f'{tag}__masks']]
        for s in range(g[f'{tag}__grad_C'].shape[0]):
            X = g[f'{tag}__step_X'][s] if m[0] else g[f'{tag}__X_0']
            sh.set_params(X, g[f'{tag}__step_C'][s], g[f'{tag}__step_Theta'][s], g[f'{tag}__step_phi_x'][s][0])
            E, gX, gC, gT, gp = _eval_all(ba, sh, nv, grad_x=m[0], grad_cam=m[1], grad_phi=m[2], **variant)
            f12_check(g, tag, s, E, gX, gC, gT, gp)
        sh.close()


def test_fit_scene_edge_cases_free_run(ba, golden):
    """The drop-in on the same inputs, free-running for the recorded iterations: energies and returned parameters."""
    g = golden('f12_edge_cases.npz')
    for tag in ('edge', 'wide', 'tiny', 'blind'):
        m = [bool(t) for t in g[f'{tag}__masks']]
        with contextlib.redirect_stdout(io.StringIO()):
            out = ba.fit_scene(g[f'{tag}__X_pix'], int(g['W']), int(g['H']), 1e-3, g[f'{tag}__E'].size, g[f'{tag}__X_0'],
                               g['C_0'], g['Theta_0'], float(g['phi_x_0']), X_mask=m[0], C_Theta_mask=m[1], phi_x_mask=m[2])
        names = (['X'] if m[0] else []) + (['C', 'Theta'] if m[1] else []) + (['phi_x'] if m[2] else [])
        assert rel_err(out[len(names)], g[f'{tag}__E']) < 1e-8, tag
        for n, v in zip(names, out):
            assert np.abs(np.asarray(v) - g[f'{tag}__final_{n}']).max() < 1e-8, (tag, n)


def test_eval_teacher_forced_f3_and_f6(ba, golden):
    for name in ('f3_lr_events.npz', 'f6_medium.npz'):
        g = golden(name)
        if 'obs_i' in g.files:
            i, j, uv = g['obs_i'], g['obs_j'], g['obs_xy'] + 0.5
            N, K = int(g['N']), int(g['K'])
        else:
            i, j, uv = orc.visibility_coo(g['X_pix'])
            N, K = g['X_pix'].shape[:2]
        sh = ba.engine.Shard(i, j, uv, N, K, int(g['W']), int(g['H']), DEV)
        for s in range(0, g['step_X'].shape[0], 5):
            sh.set_params(g['step_X'][s], g['step_C'][s], g['step_Theta'][s], g['step_phi_x'][s][0])
            E, gX, gC, gT, gp = _eval_all(ba, sh, i.size)
            assert rel_err(gX, g['grad_X'][s]) < TOL and rel_err(gC, g['grad_C'][s]) < TOL, (name, s)
            assert rel_err(gT, g['grad_Theta'][s]) < TOL and rel_err(gp, g['grad_phi_x'][s]) < TOL, (name, s)
            # row by row while the fit is far from its optimum; close to it every gradient row is a difference of
            # terms a million times larger and two CPU evaluations of the same formulas differ by more than 1e-9
            # of a row (the oracle against the reference: 3e-9 at iteration 498 of this scene)
            if (int(g['keep'][s]) if 'keep' in g.files else s) < 60:
                assert row_err(gX, g['grad_X'][s]) < TOL and row_err(gC, g['grad_C'][s]) < TOL, (name, s)
                assert row_err(gT, g['grad_Theta'][s]) < TOL, (name, s)
        sh.close()


def test_record_formats_agree(ba):
    """Pixel centres of the form int16 + 0.5 travel as 8-byte records, anything else as pack word +
    two doubles; both must give the oracle's numbers (negative pixels, the int16 limits and a
    non-representable pixel included)."""
    rng = np.random.default_rng(11)
    N, K = 400, 12
    C, Th, R = ba.scenes.orbit_cameras(K, seed=5)
    mask = rng.random((N, K)) < 0.7
    mask[:, 0] = True
    i, j = np.nonzero(mask)
    i, j = i.astype(np.int32), j.astype(np.int32)
    X = rng.normal(0, 1, (N, 3)) + np.array([0, 0, 30.0])
    pix = rng.integers(-200, 4000, (i.size, 2)).astype(np.float64)
    pix[0] = (-32767, 32767)
    pix[1] = (32767, -32767)
    for label, uv, records in (('i16', pix + 0.5, None), ('i16 forced to f64', pix + 0.5, 'f64'),
                               ('quarter pixel -> f64', pix + 0.25, None),
                               ('one pixel out of int16 range -> f64', np.vstack([[40000.5, 3.5], (pix + 0.5)[1:]]), None)):
        E_ref, gX_r, gC_r, gT_r, gp_r = orc.energy_and_gradients(X, C, Th, np.array([1.1]), 1920, 1080, i, j, uv)
        if records:
            os.environ['BA_RECORDS'] = records
        try:
            sh = ba.engine.Shard(i, j, uv, N, K, 1920, 1080, DEV)
        finally:
            os.environ.pop('BA_RECORDS', None)
        sh.set_params(X, C, Th, 1.1)
        E, gX, gC, gT, gp = _eval_all(ba, sh, i.size)
        assert abs(E - E_ref) / E_ref < TOL, label
        assert rel_err(gX, gX_r) < TOL and rel_err(gC, gC_r) < TOL, label
        assert rel_err(gT, gT_r) < TOL and rel_err(gp, gp_r) < TOL, label
        sh.close()


def test_eval_ragged_and_empty_points(ba):
    """Points without observations get a zero gradient; long tracks span several waves."""
    rng = np.random.default_rng(3)
    N, K = 500, 700
    counts = rng.integers(0, 5, N)
    counts[7] = 650          # one point seen by almost every camera (spans > 10 wave steps)
    counts[8] = 0
    counts[N - 1] = 0
    C, Th, R = ba.scenes.orbit_cameras(K, seed=2)
    i = np.repeat(np.arange(N), counts).astype(np.int32)
    j = np.concatenate([np.sort(rng.choice(K, c, replace=False)) for c in counts]).astype(np.int32)
    X = rng.normal(0, 1, (N, 3)) + np.array([0, 0, 30.0])
    uv = rng.uniform(0, 1000, (i.size, 2))
    E_ref, gX_r, gC_r, gT_r, gp_r = orc.energy_and_gradients(X, C, Th, np.array([1.1]), 1920, 1080, i, j, uv)
    for reorder, force_tiles in ((True, False), (False, False), (True, True)):
        if force_tiles:     # tiles even though they do not pay off on a scene like this
            os.environ['BA_FORCE_TILES'] = '1'
        try:
            sh = ba.engine.Shard(i, j, uv, N, K, 1920, 1080, DEV, reorder=reorder)
        finally:
            os.environ.pop('BA_FORCE_TILES', None)
        sh.set_params(X, C, Th, 1.1)
        for kw in (dict(), dict(point_major=True), dict(untiled=True), dict(cam_atomics=True)):
            sh.gX.fill_(123.0)       # stale content must not survive
            E, gX, gC, gT, gp = _eval_all(ba, sh, i.size, **kw)
            assert abs(E - E_ref) / E_ref < TOL, (reorder, kw)
            assert rel_err(gX, gX_r) < TOL and rel_err(gC, gC_r) < TOL, (reorder, kw)
            assert rel_err(gT, gT_r) < TOL and rel_err(gp, gp_r) < TOL, (reorder, kw)
            assert np.all(gX[counts == 0] == 0.0), (reorder, kw)
        sh.close()


def test_problem_create_rejects_bad_input(ba):
    i = np.array([0, 2, 1], dtype=np.int32)
    j = np.array([0, 1, 2], dtype=np.int32)
    uv = np.zeros((3, 2))
    with pytest.raises(ba.engine._lib.BAError, match='sorted'):
        ba.engine.Shard(i, j, uv, 3, 3, 10, 10, DEV, reorder=False)
    with pytest.raises(ba.engine._lib.BAError, match='out of range'):
        ba.engine.Shard(np.array([0, 1, 5], dtype=np.int32), j, uv, 3, 3, 10, 10, DEV)


# ---- H2 / H10: gauge fix and update rules ----------------------------------------------------
def test_gauge_fix_matches_reference_functions(ba, golden):
    g = golden('f5_gauge_funcs.npz')
    N, K = g['X'].shape[0], g['C'].shape[0]
    i = np.arange(N, dtype=np.int32)
    sh = ba.engine.Shard(i, np.zeros(N, dtype=np.int32), np.zeros((N, 2)), N, K, 10, 10, DEV)
    sh.set_params(g['X'], g['C'], g['Theta'], 1.0)
    sh.gauge_fix()
    assert np.abs(sh.X.cpu().numpy() - g['norm_X']).max() < 1e-13
    assert np.abs(sh.C.cpu().numpy() - g['norm_C']).max() < 1e-13
    assert np.abs(sh.Theta.cpu().numpy() - g['align_Theta']).max() < 1e-13
    assert np.abs(sh.R_gauge.cpu().numpy().reshape(K, 3, 3) - g['align_R']).max() < 1e-13
    gauge = sh.gauge.cpu().numpy()
    assert int(gauge[13]) == int(g['norm_j1'])
    sh.close()


def test_gauge_drop_ins_match_reference_functions(ba, golden):
    """ryxz, get_rotations, align_scene, normalize_scene as stand-alone drop-ins (p2w:138-140, 254-333)
    against the reference's own outputs, and the argument forms the goldens do not cover against the oracle."""
    g = golden('f5_gauge_funcs.npz')
    p = ba.pix2world
    assert np.abs(p.ryxz(g['Theta']) - g['R']).max() < 1e-14
    assert np.abs(p.get_rotations(g['R']) - g['Theta_back']).max() < 1e-13
    Xa, Ca, Ta, Ra = p.align_scene(g['X'], g['C'], Theta=g['Theta'], ret_R=True)
    for got, key in ((Xa, 'align_X'), (Ca, 'align_C'), (Ta, 'align_Theta'), (Ra, 'align_R')):
        assert np.abs(got - g[key]).max() < 1e-13, key
    assert len(p.align_scene(g['X'], g['C'], Theta=g['Theta'])) == 3
    Xn, Cn, j1 = p.normalize_scene(g['align_X'], g['align_C'], ret_j=True)
    assert int(j1) == int(g['norm_j1'])
    assert np.abs(Xn - g['norm_X']).max() < 1e-13 and np.abs(Cn - g['norm_C']).max() < 1e-13
    # rotation matrices instead of angles; chosen baseline cameras and scale
    want = orc.align_scene(g['X'], g['C'], R=g['R'], ret_R=True)
    got = p.align_scene(g['X'], g['C'], R=g['R'], ret_R=True)
    for a, b in zip(got, want):
        assert np.abs(a - b).max() < 1e-13
    want = orc.normalize_scene(g['X'], g['C'], main_indexes=(2, 5), dist_scale=3.0)
    got = p.normalize_scene(g['X'], g['C'], main_indexes=(2, 5), dist_scale=3.0)
    assert len(got) == 2 and np.abs(got[0] - want[0]).max() < 1e-13 and np.abs(got[1] - want[1]).max() < 1e-13
    assert abs(np.linalg.norm(got[1][5] - got[1][2]) - 3.0) < 1e-13
    with pytest.raises(ValueError, match='Expected Theta or R to be passed'):
        p.align_scene(g['X'], g['C'])
    with pytest.raises(ValueError, match='not both'):
        p.align_scene(g['X'], g['C'], Theta=g['Theta'], R=g['R'])
    with pytest.raises(ValueError, match='main_indexes to have length 2'):
        p.normalize_scene(g['X'], g['C'], main_indexes=(1, 2, 3))
    # inputs are left alone
    X_keep = g['X'].copy()
    p.align_scene(X_keep, g['C'], Theta=g['Theta'])
    assert np.array_equal(X_keep, g['X'])


def test_adam_and_sgd_match_reference_functions(ba, golden):
    import torch
    g = golden('f5_gauge_funcs.npz')
    N = g['adam_x'].shape[0]
    sh = ba.engine.Shard(np.arange(N, dtype=np.int32), np.zeros(N, dtype=np.int32), np.zeros((N, 2)), N, 2, 10,
                         10, DEV)
    t = lambda a: torch.from_numpy(a.copy()).to(DEV).view(-1)
    x, s, r = t(g['adam_x']), t(g['adam_s']), t(g['adam_r'])
    grad = t(g['adam_g']) * sh.nV            # the kernel multiplies by 1/|V|
    sh.adam(x, grad, s, r, 1e-2, 1.0 - 0.9 ** 3, 1.0 - 0.999 ** 3)
    assert np.abs(x.cpu().numpy() - g['adam_x_new'].ravel()).max() < 1e-14
    assert np.abs(s.cpu().numpy() - g['adam_s_new'].ravel()).max() < 1e-15
    assert np.abs(r.cpu().numpy() - g['adam_r_new'].ravel()).max() < 1e-15
    x2 = t(g['adam_x'])
    sh.sgd(x2, grad, 1e-2)
    assert np.abs(x2.cpu().numpy() - g['sgd_x_new'].ravel()).max() < 1e-15
    sh.close()


# ---- whole fit_scene -------------------------------------------------------------------------
@pytest.mark.parametrize('opt', ['Adam', 'SGD'])
def test_fit_scene_dense_api_all_masks(ba, golden, opt):
    g = golden('f1_tiny_masks.npz')
    lr = 1e-3 if opt == 'Adam' else 1e-7
    for m in g['combos']:
        m = tuple(bool(t) for t in m)
        tag = f"{opt}_{''.join('1' if t else '0' for t in m)}"
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            out = ba.fit_scene(g['X_pix'], int(g['W']), int(g['H']), lr, 5, g['X_0'], g['C_0'], g['Theta_0'],
                               float(g['phi_x_0']), X_mask=m[0], C_Theta_mask=m[1], phi_x_mask=m[2],
                               optimizer=opt, ret_all_iters=True)
        names = (['X'] if m[0] else []) + (['C', 'Theta'] if m[1] else []) + (['phi_x'] if m[2] else [])
        assert len(out) == 2 * len(names) + 1
        assert rel_err(out[len(names)], g[f'{tag}__E']) < TOL, tag
        for k, n in enumerate(names):
            assert np.abs(np.asarray(out[k]) - g[f'{tag}__final_{n}']).max() < 1e-9, (tag, n)
            hist = out[len(names) + 1 + k]
            assert hist.shape[0] == 5
            want = g[f'{tag}__step_{n}']
            got = np.asarray(hist[:want.shape[0]], dtype=np.float64).reshape(want.shape)
            assert np.abs(got - want).max() < 1e-9, (tag, n)
        if m[2]:
            assert isinstance(out[len(names) - 1], np.float64)
        assert logs_match(buf.getvalue().splitlines(), g[f'{tag}__log'].tolist()), tag


def test_fit_scene_moscow_free_run_300(ba, golden):
    g = golden('f2_moscow_ride.npz')
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        X, C, Th, phi, E = ba.fit_scene(_dense_from_coo(g), int(g['W']), int(g['H']), 1e-3, 300, g['free__X_0'],
                                        g['C_0'], g['Theta_0'], float(g['phi_x_0']), patience=1000)
    assert E.shape == g['free__E'].shape
    assert rel_err(E, g['free__E']) < 1e-8
    assert np.abs(X - g['free__final_X']).max() < 1e-6
    assert np.abs(C - g['free__final_C']).max() < 1e-6
    assert np.abs(Th - g['free__final_Theta']).max() < 1e-6
    assert abs(phi - float(g['free__final_phi_x'])) < 1e-6
    assert logs_match(buf.getvalue().splitlines(), g['free__log'].tolist())


def test_fit_scene_lr_events(ba, golden):
    g = golden('f3_lr_events.npz')
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = ba.fit_scene(g['X_pix'], int(g['W']), int(g['H']), 1e-3, 200, g['X_0'], g['C_0'], g['Theta_0'],
                           float(g['phi_x_0']), patience=20)
    events = lambda lines: [ln for ln in lines if ('LR' in ln or 'Reached' in ln)]
    assert events(buf.getvalue().splitlines()) == events(g['log'].tolist())
    assert out[-1].shape == g['E'].shape
    assert rel_err(out[-1][:100], g['E'][:100]) < 1e-8


def test_fit_scene_early_stops(ba, golden):
    """The early exits of the loop (p2w:625-638) through the dense drop-in: every printed line, the length of
    E and the returned parameters as the reference gave them (stop_value; stop_diff after rollbacks, Adam
    and SGD; lr == 0.0, which returns the rolled-back parameters)."""
    from test_oracle_golden import _f10_kwargs
    g = golden('f10_early_stops.npz')
    for tag in g['runs'].tolist():
        kw = _f10_kwargs(g, tag)
        lr, max_iters = kw.pop('lr'), int(kw.pop('max_iters'))
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            out = ba.fit_scene(g['X_pix'], int(g['W']), int(g['H']), lr, max_iters, g['X_0'], g['C_0'], g['Theta_0'],
                               float(g['phi_x_0']), **kw)
        assert logs_match(buf.getvalue().splitlines(), g[f'{tag}__log'].tolist()), (tag, buf.getvalue())
        assert out[-1].shape == g[f'{tag}__E'].shape, tag
        assert rel_err(out[-1], g[f'{tag}__E']) < 1e-8, tag
        for n, v in zip(('X', 'C', 'Theta', 'phi_x'), out):
            assert np.abs(np.asarray(v) - g[f'{tag}__final_{n}']).max() < 1e-8, (tag, n)


def test_registered_callable_optimizers(ba, golden):
    """Callables in gd_func_dict (the reference's extension point, p2w:362, 705) drive the GPU loop."""
    from test_host_cpu import check_registered_callables

    def run(g, optimizer, lr, max_iters, m, **kw):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            out = ba.fit_scene(g['X_pix'], int(g['W']), int(g['H']), lr, max_iters, g['X_0'], g['C_0'], g['Theta_0'],
                               float(g['phi_x_0']), X_mask=m[0], C_Theta_mask=m[1], phi_x_mask=m[2],
                               optimizer=optimizer, **kw)
        names = (['X'] if m[0] else []) + (['C', 'Theta'] if m[1] else []) + (['phi_x'] if m[2] else [])
        return out[len(names)], dict(zip(names, out)), buf.getvalue().splitlines()
    check_registered_callables(golden, run)


def test_queued_updates_do_not_change_the_trajectory(ba, golden):
    """The loop queues each update before it knows the energy and drops it when the LR schedule
    intervenes; it must walk the same trajectory as the run that waits every time (up to the
    summation-order noise of the floating-point atomics, which two identical runs also show)."""
    g = golden('f3_lr_events.npz')
    oi, oj, uv, _ = ba.engine.visibility_coo(g['X_pix'], DEV)
    N, K = g['X_pix'].shape[:2]
    runs = []
    for spec in (True, False):
        sh = ba.engine.Shard(oi, oj, uv, N, K, int(g['W']), int(g['H']), DEV)
        sh.set_params(g['X_0'], g['C_0'], g['Theta_0'], float(g['phi_x_0']))
        fit = ba.SceneFit(sh, 1e-3, 200, patience=20, printer=lambda *a: None, speculate=spec).run()
        runs.append((fit.energies(), sh.X.cpu().numpy(), sh.cam.cpu().numpy(), fit.dropped))
        sh.close()
    assert runs[0][3] > 0 and runs[1][3] == 0          # the schedule did overrule queued updates
    assert runs[0][0].shape == runs[1][0].shape
    assert rel_err(runs[0][0][:100], runs[1][0][:100]) < 1e-9
    assert rel_err(runs[0][0], g['E']) < 1e-6


def test_step_update_in_place_equals_out_of_place(ba, golden):
    import torch
    g = golden('f6_medium.npz')
    N, K = int(g['N']), int(g['K'])
    sh = ba.engine.Shard(g['obs_i'], g['obs_j'], g['obs_xy'] + 0.5, N, K, int(g['W']), int(g['H']), DEV)
    sh.set_params(g['X_0'], g['C_0'], g['Theta_0'], float(g['phi_x_0']))
    sh.gauge_fix()
    X0, cam0 = sh.X.clone(), sh.cam.clone()
    mx = (torch.full_like(sh.X, 1e-3), torch.full_like(sh.X, 1e-5))
    mc = (torch.full_like(sh.cam, 1e-3), torch.full_like(sh.cam, 1e-5))
    sh.eval()
    assert sh.wait_energy() == sh.energy()
    args = (2, 1e-3, 0.1, 0.001, True, (0, 6 * K + 1))
    X, cam, mx2, mc2 = sh.step_update(*args, mx, mc, True)
    assert torch.equal(sh.X, X0) and torch.equal(sh.cam, cam0)          # inputs untouched
    assert float(mx[0].max()) == 1e-3 and float(mc[1].min()) == 1e-5
    assert not torch.equal(X, X0)
    Xi, cami, mxi, mci = sh.step_update(*args, mx, mc, True, in_place=True)
    assert Xi is sh.X and cami is sh.cam and mxi[0] is mx[0] and mci[1] is mc[1]
    for a, b in ((X, Xi), (cam, cami), (mx2[0], mxi[0]), (mx2[1], mxi[1]), (mc2[0], mci[0]), (mc2[1], mci[1])):
        assert torch.equal(a, b)
    sh.close()


def test_fused_update_equals_separate_steps(ba, golden):
    """Pass with the X update fused in + camera step + pending gauge record, against
    pass, camera step, points step: three consecutive iterations."""
    import torch
    g = golden('f6_medium.npz')
    N, K = int(g['N']), int(g['K'])
    states = []
    for mode in ('merged', 'fused', 'separate'):
        fused = mode != 'separate'
        sh = ba.engine.Shard(g['obs_i'], g['obs_j'], g['obs_xy'] + 0.5, N, K, int(g['W']), int(g['H']), DEV)
        assert sh.can_fuse
        sh.set_params(g['X_0'], g['C_0'], g['Theta_0'], float(g['phi_x_0']))
        sh.gauge_fix()
        mx = (torch.zeros_like(sh.X), torch.zeros_like(sh.X))
        mc = (torch.zeros_like(sh.cam), torch.zeros_like(sh.cam))
        E = []
        for t in range(1, 4):
            bc1, bc2 = 1 - 0.9 ** t, 1 - 0.999 ** t
            if fused:
                merged = mode == 'merged'       # the camera step also finishes the pass
                X, mx = sh.eval_update(2, 1e-3, bc1, bc2, mx, defer_finalize=merged)
                cam, mc, rec = sh.camera_update(2, 1e-3, bc1, bc2, (0, 6 * K + 1), mc, True, finalize=merged)
                E.append(sh.wait_energy())
                assert E[-1] == sh.energy()
                sh.adopt(X, cam, rec)
            else:
                sh.eval()
                E.append(sh.wait_energy())
                X, cam, mx, mc = sh.step_update(2, 1e-3, bc1, bc2, True, (0, 6 * K + 1), mx, mc, True)
                sh.adopt(X, cam)
        if fused:
            assert sh._X_gauge is not None          # the last gauge record is still pending
        states.append([E] + [t.cpu().numpy() for t in (sh.X, sh.cam, mx[0], mx[1], mc[0], mc[1])])
        assert sh._X_gauge is None
        sh.close()
    for other in states[:2]:
        assert rel_err(np.array(other[0]), np.array(states[2][0])) < 1e-12
        for a, b in zip(other[1:], states[2][1:]):
            assert np.abs(a - b).max() <= 1e-11 * max(1.0, np.abs(b).max())


def test_multi_gpu_code_path_on_one_rank(ba, golden):
    """The sequence a rank runs with several GPUs (pass into the shared sum buffer, RCCL all-reduce
    of it, camera step that finishes the pass) on a one-rank RCCL group: same energies as the
    one-GPU sequence without the buffer and the collective."""
    import torch.distributed as dist
    g = golden('f6_medium.npz')
    N, K = int(g['N']), int(g['K'])
    own_group = not dist.is_initialized()
    if own_group:
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29871', rank=0, world_size=1,
                                device_id=__import__('torch').device(DEV))
    try:
        runs = []
        for force in (True, False):
            sh = ba.engine.Shard(g['obs_i'], g['obs_j'], g['obs_xy'] + 0.5, N, K, int(g['W']), int(g['H']), DEV)
            sh.distributed = force
            sh.set_params(g['X_0'], g['C_0'], g['Theta_0'], float(g['phi_x_0']))
            fit = ba.SceneFit(sh, 1e-3, 60, patience=100, printer=lambda *a: None).run()
            assert fit.fuse
            runs.append(fit.energies())
            sh.close()
        assert rel_err(runs[0], runs[1]) < 1e-9
        assert rel_err(runs[0], g['E'][:60]) < 1e-8
    finally:
        if own_group:
            dist.destroy_process_group()


def test_fit_scene_errors_match_reference_messages(ba, golden):
    g = golden('f1_tiny_masks.npz')
    args = [g['X_pix'], int(g['W']), int(g['H']), 1e-3, 5, g['X_0'], g['C_0'], g['Theta_0'], float(g['phi_x_0'])]
    with pytest.raises(ValueError, match='At least one of X_mask'):
        ba.fit_scene(*args, X_mask=False, C_Theta_mask=False, phi_x_mask=False)
    with pytest.raises(ValueError, match='Unkown name of the optmizer'):
        ba.fit_scene(*args, optimizer='LBFGS')
    bad = list(args)
    bad[8] = 4.0
    with pytest.raises(ValueError, match=r'within \(0.0, pi\)'):
        ba.fit_scene(*bad)
    few = g['X_pix'].copy()
    few[:, 1:] = -1
    a2 = list(args)
    a2[0] = few
    with pytest.raises(ValueError, match='number of equations is smaller'):
        ba.fit_scene(*a2)
    split = g['X_pix'].copy()
    split[:30, :7] = -1
    split[30:, 7:] = -1
    a3 = list(args)
    a3[0] = split
    with pytest.raises(ValueError, match='connected components of visibility graph to be 1, got 2'):
        ba.fit_scene(*a3)


# ---- scoring helpers -------------------------------------------------------------------------
def test_transform_and_error(ba, golden):
    g = golden('f5_gauge_funcs.npz')
    W, H, phi = int(g['tr_W']), int(g['tr_H']), float(g['tr_phi'])
    args = (g['tr_X'], g['tr_C'], g['Theta'], phi, W, H)
    for kw, key in ((dict(), 'tr_default'), (dict(delete_boundary_points=True), 'tr_boundary'),
                    (dict(delete_back_points=False), 'tr_keep_back')):
        got, want = ba.transform(*args, **kw), g[key]
        assert got.shape == want.shape
        assert np.array_equal(np.isnan(got), np.isnan(want)), key
        assert np.allclose(got, want, rtol=1e-12, atol=1e-9, equal_nan=True), key
    assert ba.error(g['err_X_pix'], g['tr_default']) == pytest.approx(float(g['err_value']), rel=1e-12)


# ---- linear initialisation steps (p2w:1655-1808) ------------------------------------------------
def test_triangulation_and_resection_match_reference(ba, golden):
    """get_X_from_C_Theta / get_C_Theta_from_X through the dense drop-in signatures vs the
    reference's outputs, to 1e-9.  The device forms A^T A like the reference's large-input path
    (p2w:1789-1792) while the goldens come from its SVD path; the cyclic Jacobi iteration keeps the
    relative accuracy of the small eigenpairs (numpy.linalg.eigh on the same matrix is off by 1e-5
    in pixel units), so both agree."""
    g = golden('f7_linear_init.npz')
    N, K, W, H, phi = int(g['N']), int(g['K']), int(g['W']), int(g['H']), float(g['phi_x'])
    X_pix = np.full((N, K, 2), -1, dtype=np.int32)
    X_pix[g['obs_i'], g['obs_j']] = g['obs_xy']
    for tag in ('true', 'noisy'):
        for norm, n in ((False, 'pix'), (True, 'norm')):
            X, C, Th = g[f'X_{tag}'], g[f'C_{tag}'], g[f'Theta_{tag}']
            got = ba.get_X_from_C_Theta(X_pix, C, Th, phi, W, H, normalize=norm)
            want = g[f'tri_{tag}_{n}']
            assert np.array_equal(np.isnan(got), np.isnan(want))
            assert np.nanmax(np.abs(got - want)) < 1e-9
            Cg, Tg = ba.get_C_Theta_from_X(X_pix, X, phi, W, H, normalize=norm)
            tol = 1e-9
            assert np.array_equal(np.isnan(Cg), np.isnan(g[f'res_C_{tag}_{n}']))
            assert np.nanmax(np.abs(Cg - g[f'res_C_{tag}_{n}'])) < tol
            assert np.nanmax(np.abs(Tg - g[f'res_Theta_{tag}_{n}'])) < tol
    Cg, Tg = ba.get_C_Theta_from_X(X_pix, g['X_with_nan'], phi, W, H, normalize=True, _validate=False)
    assert np.nanmax(np.abs(Cg - g['res_C_nan_norm'])) < 1e-9 and np.nanmax(np.abs(Tg - g['res_Theta_nan_norm'])) < 1e-9
    few = X_pix.copy()
    few[5:, :] = -1
    with pytest.raises(ValueError, match='at least 6 visible points'):
        ba.get_C_Theta_from_X(few, g['X_true'], phi, W, H)
    with pytest.warns(RuntimeWarning, match='at least 2 times'):
        once = X_pix.copy()
        once[:, 1:] = -1
        assert np.isnan(ba.get_X_from_C_Theta(once, g['C_true'], g['Theta_true'], phi, W, H)).all()


def test_triangulation_and_resection_vs_oracle_and_round_trip(ba):
    """A larger sparse scene: device vs the CPU restatement, and the round trip - points
    triangulated from the true cameras reproject onto their pixels to within the half-pixel
    quantisation, cameras resected from the true points land on the true cameras."""
    sc = ba.scenes.orbit_scene(N=3000, K=60, track_len=12, seed=3)
    N, K, W, H = sc['N_local'], sc['K'], sc['W'], sc['H']
    i, j, uv = sc['i'], sc['j'], sc['uv']
    Xt, Ct, Tt, phi = sc['X'], sc['C'], sc['Theta'], sc['phi_x']
    for norm in (False, True):
        Xd = ba.engine.triangulate(i, j, uv, N, Ct, Tt, phi, W, H, norm, DEV).cpu().numpy()
        Xo = orc.triangulate_obs(i, j, uv, N, Ct, Tt, phi, W, H, norm)
        assert np.array_equal(np.isnan(Xd), np.isnan(Xo))
        assert np.nanmax(np.abs(Xd - Xo)) < 1e-9
        Cd, Td = (t.cpu().numpy() for t in ba.engine.resect(i, j, uv, K, Xt, phi, W, H, norm, DEV))
        Co, To = orc.resect_obs(i, j, uv, K, Xt, phi, W, H, norm, use_gram=False)     # the SVD of the rows
        tol = 1e-9
        assert np.array_equal(np.isnan(Cd), np.isnan(Co))
        assert np.nanmax(np.abs(Cd - Co)) < tol and np.nanmax(np.abs(Td - To)) < tol
    ok = ~np.isnan(Xd[:, 0])
    assert ok.mean() > 0.95
    model = orc.transform_obs(np.nan_to_num(Xd), Ct, Tt, phi, W, H, i, j)
    vis = ok[i]
    assert np.abs(model[vis] - uv[vis]).max() < 2.0            # pixel quantisation of the tracks
    assert np.nanmax(np.abs(Cd - Ct)) < 5e-2 and np.nanmax(np.abs(Td - Tt)) < 5e-3


def test_triangulation_and_resection_on_long_tracks(ba, golden):
    """The recorded scene (93 points followed through up to 196 of 327 frames, cameras that see 10 to 48 of them):
    both forms of the triangulation kernel (a caller's list: four lanes per point; a shard's own list: two lanes) take
    tracks of many trips, the resection kernel cameras with a few points only - against the CPU
    restatement, which the reference's outputs pin on the small scene of golden F7."""
    g = golden('f2_moscow_ride.npz')
    N, K, W, H = int(g['N']), int(g['K']), int(g['W']), int(g['H'])
    i, j, uv = g['obs_i'], g['obs_j'], g['obs_xy'] + 0.5
    C_, Th, phi, X = g['free__final_C'], g['free__final_Theta'], float(g['free__final_phi_x']), g['free__final_X']
    assert np.bincount(i, minlength=N).max() > 150
    for norm in (False, True):
        want = orc.triangulate_obs(i, j, uv, N, C_, Th, phi, W, H, norm)
        got = ba.engine.triangulate(i, j, uv, N, C_, Th, phi, W, H, norm, DEV).cpu().numpy()
        sh = ba.engine.Shard(i, j, uv, N, K, W, H, DEV)
        sh.set_params(X, C_, Th, phi)
        own = sh.triangulate(normalize=norm).cpu().numpy()
        sh.close()
        for have in (got, own):
            assert np.array_equal(np.isnan(have), np.isnan(want))
            assert np.nanmax(np.abs(have - want) / (1.0 + np.abs(want))) < 1e-9
        Cd, Td = (t.cpu().numpy() for t in ba.engine.resect(i, j, uv, K, X, phi, W, H, norm, DEV))
        Co, To = orc.resect_obs(i, j, uv, K, X, phi, W, H, norm, use_gram=False)
        assert np.array_equal(np.isnan(Cd), np.isnan(Co)) and not np.isnan(Co).any()
        assert np.nanmax(np.abs(Cd - Co) / (1.0 + np.abs(Co))) < 1e-7 and np.nanmax(np.abs(Td - To)) < 1e-7


# ---- full-size properties (N = 1 M points, K = 2 k cameras, ~10 M observations) ---------------
def test_shard_triangulates_its_own_list(ba, golden):
    """Shard.triangulate (the kernel on the engine's own point order, two lanes per point) gives what the
    reference's get_X_from_C_Theta gives (golden F7, NaN pattern included) and what the stand-alone call gives on a
    scene whose points come in random order."""
    import torch
    g = golden('f7_linear_init.npz')
    N, K = int(g['N']), int(g['K'])
    for tag, norm in (('true', False), ('noisy', True), ('noisy', False)):
        sh = ba.engine.Shard(g['obs_i'], g['obs_j'], g['obs_xy'] + 0.5, N, K, int(g['W']), int(g['H']), DEV)
        sh.set_params(np.zeros((N, 3)), g[f'C_{tag}'], g[f'Theta_{tag}'], float(g['phi_x']))
        got = sh.triangulate(normalize=norm).cpu().numpy()
        want = g[f"tri_{tag}_{'norm' if norm else 'pix'}"]
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert np.nanmax(np.abs(got - want)) < 1e-9
        sh.close()
    sc = ba.scenes.orbit_scene(N=30000, K=400, track_len=9, seed=6)
    sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], 400, sc['W'], sc['H'], DEV)
    assert sh.perm is not None
    sh.set_params(sc['X_0'], sc['C'], sc['Theta'], sc['phi_x'])
    got = sh.triangulate()
    want = ba.engine.triangulate(sc['i'], sc['j'], sc['uv'], sc['N_local'], sc['C'], sc['Theta'], sc['phi_x'], sc['W'], sc['H'],
                                 False, DEV)
    assert torch.equal(torch.isnan(got), torch.isnan(want))
    assert float((got - want).abs().nan_to_num().max()) < 1e-9
    assert float((got - torch.from_numpy(sc['X']).to(DEV)).abs().nan_to_num().max()) < 0.5      # pixel quantisation only
    sh.close()


def test_full_size_invariants(ba):
    import torch
    sc = ba.scenes.orbit_scene(N=1_000_000, K=2000, track_len=10, seed=0)
    nv = sc['i'].size
    assert nv > 9_000_000
    sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], sc['K'], sc['W'], sc['H'], DEV)
    sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    sh.eval()
    E1 = sh.energy()
    gX, red = sh.gX.clone(), sh.red.clone()
    # translation invariance of the energy: sum_i dE/dX_i + sum_j dE/dC_j = 0
    tot = gX.sum(0) + sh.gC.sum(0)
    assert float(tot.abs().max()) < 1e-7 * float(gX.abs().sum())
    # the three forms of the pass agree (different summation orders only)
    for kw in (dict(point_major=True), dict(untiled=True), dict(cam_atomics=True)):
        sh.eval(**kw)
        assert abs(sh.energy() - E1) / E1 < 1e-12
        assert float((sh.red - red).abs().max()) <= 1e-10 * float(red.abs().max())
        assert float((sh.gX - gX).abs().max()) <= 1e-12 * float(gX.abs().max())
    # a sample of points and cameras against the oracle restricted to their observations
    pts = np.array([0, 1, 77777, 500000, 999999])
    sel = np.isin(sc['i'], pts)
    _, gX_r, _, _, _ = orc.energy_and_gradients(sc['X_0'], sc['C_0'], sc['Theta_0'], np.array([sc['phi_x_0']]),
                                                sc['W'], sc['H'], sc['i'][sel], sc['j'][sel], sc['uv'][sel],
                                                n_visible=nv)
    assert rel_err(sh.points_to_caller(gX).cpu().numpy()[pts] / nv, gX_r[pts]) < TOL
    cams = np.array([0, 1, 999, 1999])
    selc = np.isin(sc['j'], cams)
    _, _, gC_r, gT_r, _ = orc.energy_and_gradients(sc['X_0'], sc['C_0'], sc['Theta_0'], np.array([sc['phi_x_0']]),
                                                   sc['W'], sc['H'], sc['i'][selc], sc['j'][selc], sc['uv'][selc],
                                                   n_visible=nv)
    assert rel_err(sh.gC.cpu().numpy()[cams] / nv, gC_r[cams]) < TOL
    assert rel_err(sh.gTheta.cpu().numpy()[cams] / nv, gT_r[cams]) < TOL
    # the gauge fix leaves the energy unchanged (similarity invariance)
    sh.gauge_fix()
    sh.eval()
    assert abs(sh.energy() - E1) / E1 < 1e-9
    sh.close()


# ---- get_H_diff_matrix (p2w:1945-1995) ---------------------------------------------------------------
def test_h_diff_matrix_matches_reference(ba, golden):
    """Against the reference's own matrices.  Normalised pixels: 1e-9 everywhere.  Raw pixels: the
    columns of the DLT matrix then differ by six orders of magnitude and every solution that goes
    through A^T A loses digits: the reference's own ARPACK branch (50 common points and more) sits
    1.4e-6 from the SVD of the same rows; the device (Jacobi on A^T A) stays within 1e-6 of the SVD
    form of the oracle on every pair and of the reference's SVD branch, and within 1e-5 of its
    ARPACK branch."""
    g = golden('f9_h_diff.npz')
    X_pix = _dense_from_coo(g)
    W, H = float(g['W']), float(g['H'])
    small = g['common'] < 50
    uv = g['obs_xy'] + 0.5
    for tag, kw in (('pix', dict()), ('norm', dict(normalize=True, W=W, H=H))):
        tol = TOL if tag == 'norm' else 1e-6
        ref = g['H_diff_' + tag]
        got = ba.get_H_diff_matrix(X_pix, **kw)
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        ok = ~np.isnan(ref)
        assert np.array_equal(got, got.T, equal_nan=True)
        rel = np.where(ok, np.abs(got - ref) / np.where(ok, ref, 1.0), 0.0)
        assert rel[small].max() < tol
        assert rel.max() < (TOL if tag == 'norm' else 1e-5)
        svd = orc.h_diff_matrix_obs(g['obs_i'], g['obs_j'], uv, int(g['K']), bool(kw), W, H, use_gram=False)
        assert (np.abs(got - svd)[ok] <= tol * svd[ok]).all()
    with pytest.raises(ValueError, match='W and H have to be passed'):
        ba.get_H_diff_matrix(X_pix, normalize=True)


def test_h_diff_matrix_video_tracks_vs_oracle(ba):
    """Track-shaped visibility (a point is seen by a window of consecutive cameras), observations
    handed over in random order; against the SVD form of the oracle (normalised pixels 1e-9; raw
    pixels 1e-5: some pairs here are close to degenerate, values up to 1e8)."""
    sc = ba.scenes.orbit_scene(1500, 40, track_len=8, seed=5)
    rng = np.random.default_rng(0)
    perm = rng.permutation(sc['i'].size)
    for kw, tol in ((dict(), 1e-5), (dict(normalize=True, W=sc['W'], H=sc['H']), TOL)):
        got = ba.engine.homography_diff_matrix(sc['i'][perm], sc['j'][perm], sc['uv'][perm], sc['N'], sc['K'],
                                               device=DEV, **kw).cpu().numpy()
        ref = orc.h_diff_matrix_obs(sc['i'], sc['j'], sc['uv'], sc['K'], use_gram=False, **kw)
        assert np.array_equal(np.isnan(got), np.isnan(ref)) and (~np.isnan(ref)).sum() > 100
        ok = ~np.isnan(ref)
        assert (np.abs(got - ref)[ok] <= tol * ref[ok]).all()


def test_eighty_million_observations(ba):
    """A shard beyond the size at which torch's 2-D row gather of this ROCm build goes wrong (the
    engine reorders the observation list with it no longer): energy of the initial guess against
    the oracle's forward model, and the points come back in the caller's order."""
    sc = ba.scenes.orbit_scene(8_000_000, 2000, track_len=10, seed=0)
    sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N'], sc['K'], sc['W'], sc['H'], DEV)
    sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    sh.eval()
    E = sh.energy()
    f = sc['W'] / 2.0 / np.tan(sc['phi_x_0'] / 2.0)
    res, _, _ = orc.forward_obs(sc['X_0'], sc['C_0'], orc.ryxz(sc['Theta_0']), f, sc['W'], sc['H'],
                                sc['i'], sc['j'], sc['uv'])
    E_ref = float((res * res).sum() / (2.0 * sc['i'].size))
    assert abs(E - E_ref) <= TOL * E_ref
    assert np.array_equal(sh.points_to_caller(sh.X).cpu().numpy(), sc['X_0'])
    sh.close()


def test_many_cameras_fit_matches_oracle(ba):
    """More than 2048 cameras: the centres are scaled by a kernel of its own after the camera step
    (both the merged and the unfused form), and the camera step runs on many workgroups.  Five
    updates of fit_scene on the device against the oracle."""
    sc = ba.scenes.orbit_scene(30_000, 2500, track_len=10, seed=3)
    K = sc['K']
    ref = orc.fit_scene(sc['i'], sc['j'], sc['uv'], sc['N'], K, sc['W'], sc['H'], 1e-3, 6, sc['X_0'], sc['C_0'],
                        sc['Theta_0'], sc['phi_x_0'])
    for fuse in (True, False):
        sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N'], K, sc['W'], sc['H'], DEV)
        sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
        fit = ba.SceneFit(sh, 1e-3, 6, printer=lambda *a: None, fuse=fuse).run()
        assert fit.fuse == fuse
        assert rel_err(fit.energies(), ref[4]) < TOL
        cam = sh.cam.cpu().numpy()
        assert np.abs(sh.points_to_caller(sh.X).cpu().numpy() - ref[0]).max() < 1e-9
        assert np.abs(cam[:3 * K].reshape(K, 3) - ref[1]).max() < 1e-9
        assert np.abs(cam[3 * K:6 * K].reshape(K, 3) - ref[2]).max() < 1e-9
        assert abs(cam[6 * K] - ref[3]) < 1e-12
        sh.close()


def test_table_consumers_after_a_merged_camera_step(ba, golden):
    """The merged camera step leaves the table rows' centres unwritten; every other consumer of the
    table (point-major and chunk kernels, the second-order pass) must rebuild it on its own."""
    g = golden('f6_medium.npz')
    N, K = int(g['N']), int(g['K'])
    sh = ba.engine.Shard(g['obs_i'], g['obs_j'], g['obs_xy'] + 0.5, N, K, int(g['W']), int(g['H']), DEV)
    sh.set_params(g['X_0'], g['C_0'], g['Theta_0'], float(g['phi_x_0']))
    fit = ba.SceneFit(sh, 1e-3, 50, printer=lambda *a: None)
    fit.launch_ahead = False       # the test looks at the shard between two steps: nothing may be queued beyond them
    assert fit.fuse
    for _ in range(3):
        fit.step()
    assert sh._table_cam is sh.cam_raw                 # the next pass would reuse the table
    assert getattr(sh.cam_raw, '_ba_gauge', None) is not None      # ... and apply the pending scale of the centres
    sh.eval()                                          # camera runs: centres from the parameter vector
    E0, gC0, gX0 = sh.energy(), sh.gC.cpu().numpy().copy(), sh.gX.cpu().numpy().copy()
    for kw in (dict(point_major=True), dict(untiled=True), dict(cam_atomics=True)):
        sh._table_cam = sh.cam                         # claim the table is current, as the loop would
        sh.eval(**kw)
        assert abs(sh.energy() - E0) <= 1e-12 * E0, kw
        assert rel_err(sh.gC.cpu().numpy(), gC0) < 1e-11 and rel_err(sh.gX.cpu().numpy(), gX0) < 1e-11, kw
    for _ in range(2):
        fit.step()
    sh._table_cam = sh.cam
    sh.eval_jtj()
    gC_j = sh.gC.cpu().numpy().copy()
    sh.eval()
    assert rel_err(gC_j, sh.gC.cpu().numpy()) < 1e-11
    sh.close()


def test_parkour_shaped_scene(ba):
    """The second configuration of the benchmark list has no observation data in the reference (frames
    only); its shape - 100 keypoints over 300 frames of 3840 x 2160, tracks of up to a quarter of the
    frames - as a synthetic scene: many more cameras than points, so the tiles hold most cameras.
    Forty iterations of fit_scene against the oracle, every energy and the final parameters."""
    sc = ba.scenes.orbit_scene(100, 300, track_len=75, seed=11, W=3840, H=2160)
    N, K = sc['N'], sc['K']
    assert sc['i'].size > 2000          # about 8 % of the pairs: the field of view cuts the 75-frame windows short
    ref = orc.fit_scene(sc['i'], sc['j'], sc['uv'], N, K, sc['W'], sc['H'], 1e-3, 41, sc['X_0'], sc['C_0'],
                        sc['Theta_0'], sc['phi_x_0'])
    with contextlib.redirect_stdout(io.StringIO()):
        got = ba.fit_scene_sparse(sc['i'], sc['j'], sc['uv'], N, K, sc['W'], sc['H'], 1e-3, 41, sc['X_0'], sc['C_0'],
                                  sc['Theta_0'], sc['phi_x_0'])
    assert got[4].shape == ref[4].shape and rel_err(got[4], ref[4]) < TOL
    for a, b in zip(got[:3], ref[:3]):
        assert np.abs(a - b).max() < 1e-6
    assert abs(got[3] - ref[3]) < 1e-6


@pytest.mark.parametrize('seed', range(10))
def test_random_scene_shapes_match_oracle(ba, seed):
    """Scenes of random shape - few or many cameras per point, holes punched into the tracks, points seen once,
    cameras that see nothing, either optimiser, a short patience so that the schedule raises and lowers the learning
    rate - through the fused and the unfused loop against the oracle: every energy, the printed lines, the final
    parameters."""
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.choice([40, 300, 2500, 12000]))
    K = int(rng.choice([2, 5, 30, 150, 600]))
    track = int(rng.integers(2, max(3, min(K, 40)) + 1))
    sc = ba.scenes.orbit_scene(N, K, track_len=min(track, K), seed=seed, stride=int(rng.integers(1, 3)))
    keep = rng.random(sc['i'].size) > rng.choice([0.0, 0.1, 0.4])            # holes in the tracks
    if K > 4:
        keep &= sc['j'] != int(rng.integers(1, K))                           # a camera that sees nothing
    i, j, uv = sc['i'][keep], sc['j'][keep], sc['uv'][keep]
    if i.size < 20:
        pytest.skip('nothing left of the scene')
    opt = 'Adam' if seed % 3 else 'SGD'
    lr = 1e-3 if opt == 'Adam' else 1e-7
    kw = dict(optimizer=opt, patience=3, factor=2.0)
    iters = 25
    ref_log = []
    ref = orc.fit_scene(i, j, uv, N, K, sc['W'], sc['H'], lr, iters, sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'],
                        log=ref_log, **kw)
    for fuse in (True, False):
        lines = []
        sh = ba.engine.Shard(i, j, uv, N, K, sc['W'], sc['H'], DEV)
        sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
        fit = ba.SceneFit(sh, lr, iters, printer=lambda *a: lines.append(' '.join(str(x) for x in a)), print_step=10 ** 9,
                          fuse=fuse, **kw).run()
        assert rel_err(fit.energies(), ref[4]) < TOL, (N, K, track, opt, fuse)
        events = [ln for ln in lines if 'LR' in ln or 'Reached' in ln]
        assert events == [ln for ln in ref_log if 'LR' in ln or 'Reached' in ln]
        cam = sh.cam.cpu().numpy()
        assert np.abs(sh.points_to_caller(sh.X).cpu().numpy() - ref[0]).max() < 1e-8
        assert np.abs(cam[:3 * K].reshape(K, 3) - ref[1]).max() < 1e-8
        assert np.abs(cam[3 * K:6 * K].reshape(K, 3) - ref[2]).max() < 1e-8
        assert abs(cam[6 * K] - ref[3]) < 1e-10
        sh.close()


@pytest.mark.parametrize('N,K,track', [(50, 3000, 400), (200, 5000, 1500), (30, 1500, 900)])
def test_small_scenes_with_many_cameras(ba, N, K, track):
    """Few points seen from thousands of cameras: scenes this small always take the tile form, whatever their
    camera lists cost; five updates against the oracle."""
    sc = ba.scenes.orbit_scene(N, K, track_len=track, seed=2)
    ref = orc.fit_scene(sc['i'], sc['j'], sc['uv'], N, K, sc['W'], sc['H'], 1e-3, 6, sc['X_0'], sc['C_0'], sc['Theta_0'],
                        sc['phi_x_0'])
    sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], N, K, sc['W'], sc['H'], DEV)
    sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    fit = ba.SceneFit(sh, 1e-3, 6, printer=lambda *a: None).run()
    assert rel_err(fit.energies(), ref[4]) < TOL
    assert np.abs(sh.points_to_caller(sh.X).cpu().numpy() - ref[0]).max() < 1e-9
    sh.close()


def test_graded_tile_list(ba):
    """Tiles of unequal size (a launch's first tiles graded by BA_TILE_HEAD, its last observations cut into tiles of
    half the size): both passes and a few fused iterations against the oracle.  The grading starts at 1024 tiles, so
    the tile is made small enough for a scene the oracle handles in a second."""
    sc = ba.scenes.orbit_scene(30000, 200, track_len=8, seed=6)
    i, j, uv, N, K = sc['i'], sc['j'], sc['uv'], 30000, 200
    assert i.size > 128 * 1024
    E_ref, gX_r, gC_r, gT_r, gp_r = orc.energy_and_gradients(sc['X_0'], sc['C_0'], sc['Theta_0'], np.array([sc['phi_x_0']]),
                                                             sc['W'], sc['H'], i, j, uv)
    ref = orc.fit_scene(i, j, uv, N, K, sc['W'], sc['H'], 1e-3, 8, sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    for head, tail in (('', '0'), ('', '0.3'), ('5:64,7:96,40:128', '0.15')):
        os.environ.update(BA_TILE_OBS='128', BA_TILE_HEAD=head, BA_TILE_TAIL=tail)
        try:
            sh = ba.engine.Shard(i, j, uv, N, K, sc['W'], sc['H'], DEV)
        finally:
            for k in ('BA_TILE_OBS', 'BA_TILE_HEAD', 'BA_TILE_TAIL'):
                os.environ.pop(k, None)
        sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
        E, gX, gC, gT, gp = _eval_all(ba, sh, i.size)
        assert abs(E - E_ref) / E_ref < TOL, (head, tail)
        assert rel_err(gX, gX_r) < TOL and rel_err(gC, gC_r) < TOL and rel_err(gT, gT_r) < TOL, (head, tail)
        sh.eval_jtj()
        assert rel_err(sh.points_to_caller(sh.gX).cpu().numpy() / i.size, gX_r) < TOL, (head, tail)
        assert rel_err(sh.gC.cpu().numpy() / i.size, gC_r) < TOL, (head, tail)
        fit = ba.SceneFit(sh, 1e-3, 8, printer=lambda *a: None).run()
        assert rel_err(fit.energies(), ref[4]) < TOL, (head, tail)
        assert np.abs(sh.points_to_caller(sh.X).cpu().numpy() - ref[0]).max() < 1e-9, (head, tail)
        sh.close()
