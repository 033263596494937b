"""Second-order extension (J^T J blocks, Schur complement, fp64 Cholesky, LM step).  The reference has no
such solver, but it forms the per-observation Jacobians these quantities are made of (p2w:654-657, 694-695,
700-701): golden F11 holds them as the reference computed them at its own iterates, and
test_second_order_blocks_from_the_reference_jacobians pins V, B, bphi, hphi, the reduced system S / rhs and the
damped step to what those arrays give.  The other tests use oracle.dense_jacobian on scenes of other shapes;
tests/test_oracle_golden.py checks that restatement against the same golden entry by entry."""
import importlib
import os

import numpy as np
import pytest
import torch

from conftest import f11_case, normal_equations_from_jacobians, rel_err, row_err
from oracle import ba_oracle as orc

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _scene(ba, N=300, K=80, seed=4, tile_obs=None, env=None):
    import os
    sc = ba.scenes.orbit_scene(N=N, K=K, track_len=8, seed=seed)
    assert sc['i'].size > 4 * N
    env = dict(env or {})
    if tile_obs:
        env['BA_TILE_OBS'] = str(tile_obs)
    os.environ.update(env)
    try:
        sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], K, sc['W'], sc['H'], DEV)
    finally:
        for k in env:
            os.environ.pop(k, None)
    sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    r, J = orc.dense_jacobian(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'], sc['W'], sc['H'], sc['i'],
                              sc['j'], sc['uv'])
    return sc, sh, r, J


# every layout of the J^T J pass; the dense block layout is forced (BA_BLOCK_MIN_FILL=0) because the
# small test scenes fill their blocks poorly, once per record format and once with several tiles and
# camera batches per tile
@pytest.mark.parametrize('layout,env,tile_obs', [
    ('blocks', {'BA_BLOCK_MIN_FILL': '0'}, None),
    ('blocks', {'BA_BLOCK_MIN_FILL': '0', 'BA_RECORDS': 'f64'}, None),
    ('blocks', {'BA_BLOCK_MIN_FILL': '0'}, 256),
    ('runs', {}, None),
    ('point-major', {}, None),
], ids=['dense-blocks', 'dense-blocks-f64-records', 'dense-blocks-small-tiles', 'camera-runs', 'point-major-tiles'])
def test_jtj_blocks_match_dense_normal_equations(ba, layout, env, tile_obs):
    sc, sh, r, J = _scene(ba, env=env, tile_obs=tile_obs)
    N, K = sc['N_local'], sc['K']
    sh.eval_jtj(point_major=layout == 'point-major', dense_blocks=layout == 'blocks')
    H, g = J.T @ J, J.T @ r
    back = lambda t: sh.points_to_caller(t).cpu().numpy()
    assert abs(float(sh.E_sum.item()) - r @ r) / (r @ r) < 1e-12
    assert rel_err(back(sh.gX).ravel(), g[:3 * N]) < 1e-10
    assert rel_err(sh.gC.cpu().numpy().ravel(), g[3 * N:3 * N + 3 * K]) < 1e-10
    assert rel_err(sh.gTheta.cpu().numpy().ravel(), g[3 * N + 3 * K:3 * N + 6 * K]) < 1e-10
    assert abs(float(sh.gphi.item()) - g[-1]) < 1e-10 * abs(g[-1])
    V = back(sh.V)
    iu = np.triu_indices(3)
    V_ref = np.array([H[3 * i:3 * i + 3, 3 * i:3 * i + 3][iu] for i in range(N)])
    assert rel_err(V, V_ref) < 1e-10
    # the point x phi column is formed by the Schur build (the pass does not store cross blocks)
    sh.lm_step(1e-3)
    assert rel_err(back(sh.Xphi), H[:3 * N, -1].reshape(N, 3)) < 1e-10
    iu6 = np.triu_indices(6)
    B_ref = np.zeros((K, 21))
    b_ref = np.zeros((K, 6))
    for j in range(K):
        idx = np.r_[3 * N + 3 * j + np.arange(3), 3 * N + 3 * K + 3 * j + np.arange(3)]
        B_ref[j] = H[np.ix_(idx, idx)][iu6]
        b_ref[j] = H[idx, -1]
    assert rel_err(sh.B.cpu().numpy(), B_ref) < 1e-10
    assert rel_err(sh.bphi.cpu().numpy(), b_ref) < 1e-10
    assert abs(float(sh.hphi.item()) - H[-1, -1]) < 1e-10 * H[-1, -1]
    sh.close()


@pytest.mark.parametrize('layout', ['default', 'point-major', 'blocks'])
@pytest.mark.parametrize('which,t', [('f1', 0), ('f1', 1), ('f1', 2), ('f1', 3), ('f2', 0), ('f2', 1)])
def test_second_order_blocks_from_the_reference_jacobians(ba, golden, which, t, layout):
    """Teacher-forced at the reference's iterates (F1 scene, moscow_ride tracks): the blocks of the J^T J pass,
    the reduced camera system of the Schur build and the damped step equal what the REFERENCE's per-observation
    Jacobians (golden F11, read out of fit_scene's frame while it ran) give, to 1e-9."""
    c = f11_case(golden, which, t)
    N, K, lam = c['N'], c['K'], 1e-3
    env = {'BA_BLOCK_MIN_FILL': '0'} if layout == 'blocks' else {}
    os.environ.update(env)
    try:
        sh = ba.engine.Shard(c['i'], c['j'], c['uv'], N, K, c['W'], c['H'], DEV)
    finally:
        for k in env:
            os.environ.pop(k, None)
    sh.set_params(c['step_X'], c['step_C'], c['step_Theta'], float(c['step_phi_x'][0]))
    sh.eval_jtj(point_major=layout == 'point-major', dense_blocks=layout == 'blocks')
    ne = normal_equations_from_jacobians(c, lam)
    back = lambda x: sh.points_to_caller(x).cpu().numpy()
    rr = float((c['r'] ** 2).sum())
    assert abs(float(sh.E_sum.item()) - rr) <= 1e-9 * rr
    g = ne['g']
    gcam = g[3 * N:3 * N + 6 * K].reshape(K, 6)
    assert row_err(back(sh.gX), g[:3 * N].reshape(N, 3)) < 1e-9
    assert row_err(sh.gC.cpu().numpy(), gcam[:, :3]) < 1e-9 and row_err(sh.gTheta.cpu().numpy(), gcam[:, 3:]) < 1e-9
    assert abs(float(sh.gphi.item()) - g[-1]) <= 1e-9 * abs(g[-1])
    assert row_err(back(sh.V), ne['V']) < 1e-9
    assert row_err(sh.B.cpu().numpy(), ne['B']) < 1e-9
    assert row_err(sh.bphi.cpu().numpy(), ne['bphi']) < 1e-9
    assert abs(float(sh.hphi.item()) - ne['hphi']) <= 1e-9 * ne['hphi']
    lib, p = sh.lib, ba.engine.ptr
    ba.engine.check(lib.ba_schur_build(sh.handle, p(sh.X), lam, p(sh.V), p(sh.Xphi), p(sh.gX), p(sh.B), p(sh.bphi),
                                       p(sh.hphi), p(sh.gC), p(sh.gTheta), p(sh.gphi), 1, p(sh.S), p(sh.rhs), None))
    n = sh.n_red
    S, rhs = torch.tril(sh.S[:n, :n]).cpu().numpy(), sh.rhs[:n].cpu().numpy()
    assert np.abs(S - np.tril(ne['S'])).max() <= 1e-9 * np.abs(ne['S']).max()
    assert np.abs(rhs - ne['rhs']).max() <= 1e-9 * np.abs(ne['rhs']).max()
    assert rel_err(back(sh.Xphi), ne['Xphi']) < 1e-9
    # the damped step itself (conditioning of the reduced system: looser, like the other step tests)
    sh.S.zero_()
    assert sh.lm_step(lam) == 0
    seen = np.repeat(np.bincount(c['i'], minlength=N) > 0, 3)
    d = np.zeros(3 * N + 6 * K + 1)
    d[np.r_[seen, np.ones(6 * K + 1, bool)]] = ne['step']
    per_cam = d[3 * N:3 * N + 6 * K].reshape(K, 6)       # the step leaves as [dC | dTheta | dphi]
    assert rel_err(sh.dcam.cpu().numpy(), np.r_[per_cam[:, :3].ravel(), per_cam[:, 3:].ravel(), d[-1]]) < 1e-6
    assert rel_err(back(sh.dX).ravel()[seen], d[:3 * N][seen]) < 1e-6
    sh.close()


def test_cholesky_solve_random_spd(ba):
    lib = ba.engine._lib.load()
    import ctypes as C
    rng = np.random.default_rng(0)
    for n in (64, 200, 700, 1000, 1500, 3400):       # 3400: four outer panels
        ld = (n + 63) // 64 * 64
        A = rng.normal(size=(n, 2 * n))
        S = A @ A.T + n * np.eye(n)
        b = rng.normal(size=n)
        Sp = np.eye(ld)
        Sp[:n, :n] = S
        Sp[np.triu_indices(ld, 1)] = np.nan          # the factorisation must never read the upper triangle
        Sd = torch.from_numpy(Sp).to(DEV)
        bd = torch.zeros(ld, dtype=torch.float64, device=DEV)
        bd[:n] = torch.from_numpy(b)
        info = C.c_int(7)
        ba.engine.check(lib.ba_cholesky_solve(n, ld, C.c_void_p(Sd.data_ptr()), C.c_void_p(bd.data_ptr()),
                                              C.byref(info), None))
        assert info.value == 0
        x = bd.cpu().numpy()[:n]
        assert rel_err(x, np.linalg.solve(S, b)) < 1e-10
        L = np.tril(Sd.cpu().numpy()[:n, :n])
        assert rel_err(L, np.linalg.cholesky(S)) < 1e-11
    # not positive definite -> pivot reported
    Sbad = -np.eye(64)
    Sd = torch.from_numpy(Sbad).to(DEV)
    bd = torch.zeros(64, dtype=torch.float64, device=DEV)
    info = C.c_int(0)
    assert lib.ba_cholesky_solve(64, 64, C.c_void_p(Sd.data_ptr()), C.c_void_p(bd.data_ptr()), C.byref(info), None) == -4
    assert info.value == 1


def test_cholesky_skips_structural_zeros_correctly(ba):
    """Block-banded matrices with a dense last row (the shape of a video's reduced camera system), one of them with
    a row in the middle that reaches back to the first columns: the factorisation leaves out the rows a panel cannot
    touch and must still give the dense result - the solution always, the factor where the natural order is kept
    (a long narrow band is cut into chains that are factorised side by side, in an order of their own)."""
    lib = ba.engine._lib.load()
    import ctypes as C
    rng = np.random.default_rng(3)
    for n, bw, far_row in ((1500, 70, None), (3000, 200, None), (5000, 90, 2600), (4097, 300, 1000), (9000, 50, None),
                           (7001, 130, 3000)):
        ld = (n + 63) // 64 * 64
        S = np.zeros((n, n))
        for d in range(1, bw + 1):
            v = rng.normal(size=n - d)
            S[np.arange(d, n), np.arange(n - d)] = v
        S[n - 1, :n - 1] = rng.normal(size=n - 1)            # the arrow
        if far_row is not None:
            S[far_row, :40] = rng.normal(size=40)            # one row far outside the band
            S[n - 60:n - 1, :50] = rng.normal(size=(59, 50))  # and the last rows coupled with the first columns (closed path)
        S = S + S.T
        S[np.diag_indices(n)] = np.abs(S).sum(axis=1) + 1.0  # diagonally dominant: positive definite
        b = rng.normal(size=n)
        Sp = np.eye(ld)
        Sp[:n, :n] = S
        Sp[np.triu_indices(ld, 1)] = np.nan
        want_x, want_L = np.linalg.solve(S, b), np.linalg.cholesky(S)
        for switch in (None, 'BA_CHOL_NO_CHAINS', 'BA_CHOL_DENSE'):
            Sd = torch.from_numpy(Sp).to(DEV)
            bd = torch.zeros(ld, dtype=torch.float64, device=DEV)
            bd[:n] = torch.from_numpy(b)
            info = C.c_int(7)
            if switch:
                os.environ[switch] = '1'
            try:
                ba.engine.check(lib.ba_cholesky_solve(n, ld, C.c_void_p(Sd.data_ptr()), C.c_void_p(bd.data_ptr()),
                                                      C.byref(info), None))
            finally:
                os.environ.pop(switch or '', None)
            assert info.value == 0
            assert rel_err(bd.cpu().numpy()[:n], want_x) < 1e-10, (n, switch)
            if switch:
                assert rel_err(np.tril(Sd.cpu().numpy()[:n, :n]), want_L) < 1e-11, (n, switch)


def _banded_arrow_system(n, bw, closed):
    """A diagonally dominant band of half-width bw with a dense last row (and, if closed, the last rows coupled with
    the first columns) in lower band storage, a right-hand side, and the solution by LAPACK's banded Cholesky with
    the border eliminated by blocks: (band, arrow, corner, b, x)."""
    import scipy.linalg
    rng = np.random.default_rng(n + bw)
    # lower band storage for scipy (row d = d-th subdiagonal), plus an arrow row and, if closed, a corner block
    band = rng.normal(size=(bw + 1, n))
    for d in range(1, bw + 1):
        band[d, n - d:] = 0.0
    arrow = rng.normal(size=n - 1)
    corner = rng.normal(size=(59, 50)) if closed else None
    row_sum = np.abs(band[1:]).sum(axis=0)
    for d in range(1, bw + 1):
        row_sum[d:] += np.abs(band[d, :n - d])
    row_sum[:n - 1] += np.abs(arrow)
    row_sum[n - 1] += np.abs(arrow).sum()
    if closed:
        row_sum[n - 60:n - 1] += np.abs(corner).sum(axis=1)
        row_sum[:50] += np.abs(corner).sum(axis=0)
    band[0] = row_sum + 1.0
    b = rng.normal(size=n)
    # reference: the band by LAPACK, arrow and corner as a low-rank correction (Woodbury on S = B + U C U^T is more
    # than a test needs: solve the bordered system by block elimination instead)
    m = 60 if closed else 1                      # the last m unknowns form the border
    nb_ = n - m
    Bb = band[:, :nb_].copy()
    for d in range(1, bw + 1):
        Bb[d, max(nb_ - d, 0):] = 0.0
    F = np.zeros((nb_, m))                      # coupling of the border's unknowns with the band's
    D = np.zeros((m, m))
    for jj in range(m):
        col = nb_ + jj
        for d in range(1, bw + 1):              # band entries that cross into the border: S[col, col - d]
            if col - d < nb_ and col - d >= 0:
                F[col - d, jj] += band[d, col - d]
            elif col - d >= nb_:
                D[jj, col - d - nb_] += band[d, col - d]
        D[jj, jj] = band[0, col]
    F[:, m - 1] += arrow[:nb_]
    D[m - 1, :m - 1] += arrow[nb_:]
    if closed:
        F[:50, :59] += corner.T
    D = np.tril(D) + np.tril(D, -1).T
    cb = scipy.linalg.cholesky_banded(Bb, lower=True)
    Y = scipy.linalg.cho_solve_banded((cb, True), np.c_[F, b[:nb_]])
    Sc = D - F.T @ Y[:, :m]
    x2 = np.linalg.solve(Sc, b[nb_:] - F.T @ Y[:, m])
    want = np.r_[Y[:, m] - Y[:, :m] @ x2, x2]

    return band, arrow, corner, b, want


def _last_plan(lib):
    import ctypes as C
    buf, n_levels = (C.c_int64 * 16)(), C.c_int(0)
    assert lib.ba_cholesky_last_plan(buf, 4, C.byref(n_levels)) == 0
    return [tuple(buf[4 * k:4 * k + 4]) for k in range(n_levels.value)]


@pytest.mark.parametrize('n,bw,closed', [(12001, 60, False), (12001, 60, True), (20000, 100, True), (30001, 40, False),
                                         (30001, 40, True)])
def test_cholesky_nested_levels_of_chains(ba, n, bw, closed):
    """What lies behind the chains of a long narrow band is a band with a border again (the separators, coupled
    through the chains between them) and is cut into chains of its own: one, two and three levels give the solution
    of a banded LAPACK solve of the same system, the deeper splits really happen (ba_cholesky_last_plan), and S comes
    back cleared.  closed: the last rows are coupled with the first columns, as on a closed camera path."""
    import ctypes as C
    lib = ba.engine._lib.load()
    ld = (n + 63) // 64 * 64
    nb = ld // 64
    band, arrow, corner, b, want = _banded_arrow_system(n, bw, closed)
    Sd0 = torch.zeros((ld, ld), dtype=torch.float64, device=DEV)
    idx = torch.arange(n, device=DEV)
    for d in range(0, bw + 1):
        Sd0[idx[d:], idx[:n - d]] = torch.from_numpy(band[d, :n - d]).to(DEV)
    Sd0[n - 1, :n - 1] += torch.from_numpy(arrow).to(DEV)
    if closed:
        Sd0[n - 60:n - 1, :50] += torch.from_numpy(corner).to(DEV)
    pad = torch.arange(n, ld, device=DEV)
    Sd0[pad, pad] = 1.0
    first = np.array([max(0, (64 * ib - bw) // 64) for ib in range(nb)], dtype=np.int32)
    first[(n - 60 if closed else n - 1) // 64:] = 0
    first_d = torch.from_numpy(first).to(DEV)
    plans = {}
    for levels in (1, 2, 3):
        Sd = Sd0.clone()
        bd = torch.zeros(ld, dtype=torch.float64, device=DEV)
        bd[:n] = torch.from_numpy(b)
        info = C.c_int(7)
        os.environ['BA_CHOL_LEVELS'] = str(levels)
        try:
            ba.engine.check(lib.ba_cholesky_solve_tracked(n, ld, C.c_void_p(Sd.data_ptr()), C.c_void_p(bd.data_ptr()),
                                                          C.c_void_p(first_d.data_ptr()), 1, C.byref(info), None))
        finally:
            os.environ.pop('BA_CHOL_LEVELS', None)
        assert info.value == 0
        assert rel_err(bd.cpu().numpy()[:n], want) < 1e-10, levels
        assert float(Sd.abs().max()) == 0.0, levels
        plans[levels] = _last_plan(lib)
        del Sd
    # dependent block steps: the chains of every level, then what is left behind the last chains
    steps = lambda plan: sum(p[2] for p in plan) + nb - (plan[-1][0] + plan[-1][1] * plan[-1][2])
    assert len(plans[1]) == 1 and plans[2][0] == plans[1][0]
    if n >= 30000:          # (n = 12 001: only eleven separators are left, a second level would not pay for itself;
        assert len(plans[2]) == 2       # n = 20 000: separators of two blocks, coupled over three - too wide for chains of so few)
    if len(plans[2]) == 2:
        assert plans[2][1][0] == plans[1][0][1] * plans[1][0][2]      # the second level starts behind the first one's chains
        assert steps(plans[2]) < steps(plans[1]) - 3
    assert steps(plans[3]) <= steps(plans[2])


@pytest.mark.parametrize('clear', [0, 1])
def test_cholesky_switch_order_does_not_reuse_stale_lists(ba, clear):
    """The solver keeps the lists that steer its block moves and its clearing from call to call while the block
    profile stays the same.  The split into chains changes them under one profile: unchained first, chained second
    (and back) must each upload their own lists.  With and without clearing S on return."""
    lib = ba.engine._lib.load()
    import ctypes as C
    rng = np.random.default_rng(5)
    n, bw = 9000, 50
    ld = (n + 63) // 64 * 64
    S = np.zeros((n, n))
    for d in range(1, bw + 1):
        S[np.arange(d, n), np.arange(n - d)] = rng.normal(size=n - d)
    S[n - 1, :n - 1] = rng.normal(size=n - 1)
    S = S + S.T
    S[np.diag_indices(n)] = np.abs(S).sum(axis=1) + 1.0
    b = rng.normal(size=n)
    want_x = np.linalg.solve(S, b)
    Sp = np.zeros((ld, ld))
    Sp[:n, :n] = np.tril(S)
    Sp[np.arange(n, ld), np.arange(n, ld)] = 1.0
    nb = ld // 64
    first = np.array([max(0, (64 * ib - bw) // 64) for ib in range(nb)], dtype=np.int32)
    first[(n - 1) // 64:] = 0
    first_d = torch.from_numpy(first).to(DEV)
    for switch in ('BA_CHOL_NO_CHAINS', None, 'BA_CHOL_NO_CHAINS', None):
        Sd = torch.from_numpy(Sp).to(DEV)
        bd = torch.zeros(ld, dtype=torch.float64, device=DEV)
        bd[:n] = torch.from_numpy(b)
        info = C.c_int(7)
        if switch:
            os.environ[switch] = '1'
        try:
            ba.engine.check(lib.ba_cholesky_solve_tracked(n, ld, C.c_void_p(Sd.data_ptr()), C.c_void_p(bd.data_ptr()),
                                                          C.c_void_p(first_d.data_ptr()), clear, C.byref(info), None))
        finally:
            os.environ.pop(switch or '', None)
        assert info.value == 0
        assert rel_err(bd.cpu().numpy()[:n], want_x) < 1e-10, switch
        if clear:
            assert float(Sd.abs().max()) == 0.0, switch


def test_queued_cholesky_takes_the_last_profile_and_notices_when_it_moved(ba):
    """ba_cholesky_solve_queued waits for nothing after its first call: the verdict (ba_cholesky_queued_result) is 0
    with the right solution while the profile stays; a pivot index for a matrix that is not positive definite; -1 when
    a row starts left of the assumed profile, after which the next call reads the profile again and solves."""
    lib = ba.engine._lib.load()
    import ctypes as C
    rng = np.random.default_rng(11)
    n = 9000
    ld = (n + 63) // 64 * 64
    nb = ld // 64

    def system(bw):
        S = np.zeros((n, n))
        for d in range(1, bw + 1):
            S[np.arange(d, n), np.arange(n - d)] = rng.normal(size=n - d)
        S[n - 1, :n - 1] = rng.normal(size=n - 1)
        S = S + S.T
        S[np.diag_indices(n)] = np.abs(S).sum(axis=1) + 1.0
        Sp = np.zeros((ld, ld))
        Sp[:n, :n] = np.tril(S)
        Sp[np.arange(n, ld), np.arange(n, ld)] = 1.0
        first = np.array([max(0, (64 * ib - bw) // 64) for ib in range(nb)], dtype=np.int32)
        first[(n - 1) // 64:] = 0
        return S, Sp, torch.from_numpy(first).to(DEV)

    def solve(Sp, first_d, b):
        Sd = torch.from_numpy(Sp).to(DEV)
        bd = torch.zeros(ld, dtype=torch.float64, device=DEV)
        bd[:n] = torch.from_numpy(b)
        ba.engine.check(lib.ba_cholesky_solve_queued(n, ld, C.c_void_p(Sd.data_ptr()), C.c_void_p(bd.data_ptr()),
                                                     C.c_void_p(first_d.data_ptr()), 1, None))
        torch.cuda.synchronize()
        info = C.c_int(7)
        ba.engine.check(lib.ba_cholesky_queued_result(C.byref(info)))
        return info.value, bd.cpu().numpy()[:n], Sd

    S1, Sp1, f1 = system(50)
    S2, Sp2, f2 = system(50)            # other values, the same profile
    S3, Sp3, f3 = system(150)           # rows that start two blocks further left
    for S, Sp in ((S1, Sp1), (S2, Sp2), (S1, Sp1)):
        b = rng.normal(size=n)
        verdict, x, Sd = solve(Sp, f1, b)
        assert verdict == 0
        assert rel_err(x, np.linalg.solve(S, b)) < 1e-10
        assert float(Sd.abs().max()) == 0.0
    bad = Sp2.copy()
    bad[4000, 4000] = -1.0
    verdict, _, _ = solve(bad, f1, rng.normal(size=n))
    assert verdict > 0
    b = rng.normal(size=n)
    verdict, _, _ = solve(Sp3, f3, b)
    assert verdict == -1
    verdict, x, Sd = solve(Sp3, f3, b)
    assert verdict == 0
    assert rel_err(x, np.linalg.solve(S3, b)) < 1e-10
    assert float(Sd.abs().max()) == 0.0
    # a narrower profile than the assumed one is fine (the assumed one is a lower bound)
    b = rng.normal(size=n)
    verdict, x, _ = solve(Sp1, f1, b)
    assert verdict == 0
    assert rel_err(x, np.linalg.solve(S1, b)) < 1e-10


def test_tracked_build_and_solve_keep_S_clean(ba):
    """The pair of entry points an LM loop uses: the Schur build reports the block profile of S (a lower bound of every
    block row's first occupied block, exact here), the solver takes it instead of reading S, gives the solution of the
    plain pair, and leaves S cleared for the next build."""
    import ctypes as C
    sc, sh, r, J = _scene(ba, N=900, K=130, seed=4)
    lib, p = sh.lib, ba.engine._lib.ptr
    lam = 1e-2
    sh.eval_jtj()
    n, ld = sh.n_red, sh.ld_red

    def build(first, is_zero):
        ba.engine.check(lib.ba_schur_build_tracked(sh.handle, p(sh.X), lam, p(sh.V), p(sh.Xphi), p(sh.gX), p(sh.B),
                                                   p(sh.bphi), p(sh.hphi), p(sh.gC), p(sh.gTheta), p(sh.gphi), 1, p(sh.S),
                                                   p(sh.rhs), p(first) if first is not None else None, is_zero, None))
    first = torch.full((ld // 64,), -7, dtype=torch.int32, device=DEV)
    build(first, 0)
    S0, rhs0 = sh.S.clone(), sh.rhs.clone()
    Sl = torch.tril(S0).cpu().numpy()
    got = first.cpu().numpy()
    for ib in range(ld // 64):
        cols = np.nonzero(np.abs(Sl[64 * ib:64 * ib + 64]).sum(axis=0))[0]
        assert got[ib] <= cols[0] // 64 and got[ib] >= 0
        if 64 * ib + 63 < 6 * sc['K']:
            assert got[ib] == cols[0] // 64       # camera rows: exact
    info = C.c_int(0)
    ba.engine.check(lib.ba_cholesky_solve(n, ld, p(sh.S), p(sh.rhs), C.byref(info), None))
    want = sh.rhs[:n].clone()
    sh.S.copy_(S0)
    sh.rhs.copy_(rhs0)
    ba.engine.check(lib.ba_cholesky_solve_tracked(n, ld, p(sh.S), p(sh.rhs), p(first), 1, C.byref(info), None))
    assert info.value == 0
    assert rel_err(sh.rhs[:n].cpu().numpy(), want.cpu().numpy()) < 1e-12
    assert float(sh.S.abs().max()) == 0.0
    # the next build into the cleared S, told so, gives the same system as a build that clears S itself
    build(first, 1)
    assert torch.equal(torch.tril(sh.S), torch.tril(S0)) or rel_err(torch.tril(sh.S).cpu().numpy(), Sl) < 1e-13
    # a profile that claims more than the rows can have is refused
    bad = first.clone()
    bad[2] = 5
    assert lib.ba_cholesky_solve_tracked(n, ld, p(sh.S), p(sh.rhs), p(bad), 0, C.byref(info), None) == -1
    sh.close()


def test_lm_step_matches_dense_solve(ba):
    sc, sh, r, J = _scene(ba, N=250, K=60, seed=9)
    N, K = sc['N_local'], sc['K']
    lam = 1e-2
    sh.eval_jtj()
    assert sh.lm_step(lam) == 0
    H, g = J.T @ J, J.T @ r
    d = np.linalg.solve(H + lam * np.diag(np.diag(H)), -g)
    dX = sh.points_to_caller(sh.dX).cpu().numpy().ravel()
    dcam = sh.dcam.cpu().numpy()
    assert rel_err(dcam, d[3 * N:]) < 1e-7
    assert rel_err(dX, d[:3 * N]) < 1e-7
    sh.close()


def test_rejected_trial_goes_back_without_a_new_pass(ba):
    """The trial point comes out of the back-substitution (X + dX, cam + dcam exactly) into tensors of its own
    and is evaluated into the second set of buffers; going back (restore, swap_evaluation, rebuild_table) leaves
    the engine where a fresh evaluation of the kept point would: the next damped step is the same."""
    sc, sh, r, J = _scene(ba, N=250, K=60, seed=9)
    sh.eval_jtj()
    E0 = sh.energy()
    X0, cam0 = sh.X.clone(), sh.cam.clone()
    assert sh.lm_step(1e-2, trial=True) == 0
    assert torch.equal(sh.X_trial, X0 + sh.dX) and torch.equal(sh.cam_trial, cam0 + sh.dcam)
    assert torch.equal(sh.X, X0) and torch.equal(sh.cam, cam0)
    kept = sh.snapshot()
    sh.adopt(sh.X_trial, sh.cam_trial)
    sh.gauge_fix()
    sh.swap_evaluation()
    sh.eval_jtj()
    assert sh.energy() != E0
    sh.restore(kept)
    sh.swap_evaluation()
    sh.rebuild_table()
    assert sh.energy() == E0 and torch.equal(sh.X, X0) and torch.equal(sh.cam, cam0)
    assert sh.lm_step(4e-2) == 0
    d_back, dX_back = sh.dcam.clone(), sh.dX.clone()
    sh.invalidate_table()
    sh.eval_jtj()
    assert sh.lm_step(4e-2) == 0
    assert rel_err(d_back.cpu().numpy(), sh.dcam.cpu().numpy()) < 1e-9
    assert rel_err(dX_back.cpu().numpy(), sh.dX.cpu().numpy()) < 1e-9
    sh.close()


def test_schur_tile_kernel_matches_point_kernel_and_dense(ba):
    """Few cameras per tile: the Schur complement is summed per tile in LDS.  Same S and rhs as the
    one-wave-per-point kernel, same step as the dense normal equations."""
    import os
    sc, sh, r, J = _scene(ba, N=800, K=60, seed=12, tile_obs=256)
    N, K = sc['N_local'], sc['K']
    lam = 1e-2
    sh.eval_jtj()
    assert sh.lm_step(lam) == 0
    d_tile, dX_tile = sh.dcam.clone(), sh.dX.clone()
    lib, p = sh.lib, ba.engine.ptr
    def build():
        ba.engine.check(lib.ba_schur_build(sh.handle, p(sh.X), lam, p(sh.V), p(sh.Xphi), p(sh.gX), p(sh.B), p(sh.bphi),
                                           p(sh.hphi), p(sh.gC), p(sh.gTheta), p(sh.gphi), 1, p(sh.S), p(sh.rhs), None))
        n = sh.n_red
        return torch.tril(sh.S[:n, :n]).cpu().numpy(), sh.rhs[:n].cpu().numpy()
    S_tile, rhs_tile = build()
    os.environ['BA_SCHUR_POINTS'] = '1'
    try:
        S_pts, rhs_pts = build()
    finally:
        os.environ.pop('BA_SCHUR_POINTS', None)
    assert np.abs(S_tile - S_pts).max() <= 1e-11 * np.abs(S_pts).max()
    assert np.abs(rhs_tile - rhs_pts).max() <= 1e-11 * np.abs(rhs_pts).max()
    H, g = J.T @ J, J.T @ r
    d = np.linalg.solve(H + lam * np.diag(np.diag(H)), -g)
    assert rel_err(d_tile.cpu().numpy(), d[3 * N:]) < 1e-7
    assert rel_err(sh.points_to_caller(dX_tile).cpu().numpy().ravel(), d[:3 * N]) < 1e-7
    sh.close()


def test_schur_tile_kernel_mixed_track_lengths(ba):
    """Tracks of 0 to 14 observations side by side: the tile kernel takes two points per trip when neither has more than
    ten observations and one at a time otherwise; a point nobody sees contributes nothing.  Same S, rhs and
    point x phi column as the one-wave-per-point kernel, and as the dense normal equations."""
    import os
    sc = ba.scenes.orbit_scene(N=500, K=300, track_len=14, seed=3)
    rng = np.random.default_rng(8)
    keep = rng.random(sc['i'].size) < rng.choice([1.0, 0.8, 0.45, 0.0], size=500, p=[0.4, 0.3, 0.25, 0.05])[sc['i']]
    keep |= sc['i'] < 3          # the first points keep all 14
    i, j, uv = sc['i'][keep], sc['j'][keep], sc['uv'][keep]
    deg = np.bincount(i, minlength=500)
    assert deg.max() == 14 and (deg == 0).any() and ((deg > 0) & (deg <= 10)).any() and (deg > 10).sum() > 20
    N, K = 500, 300
    os.environ['BA_TILE_OBS'] = '256'
    try:
        sh = ba.engine.Shard(i, j, uv, N, K, sc['W'], sc['H'], DEV)
    finally:
        os.environ.pop('BA_TILE_OBS', None)
    sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    sh.eval_jtj()
    lam = 1e-2
    lib, p = sh.lib, ba.engine.ptr
    def build():
        ba.engine.check(lib.ba_schur_build(sh.handle, p(sh.X), lam, p(sh.V), p(sh.Xphi), p(sh.gX), p(sh.B), p(sh.bphi),
                                           p(sh.hphi), p(sh.gC), p(sh.gTheta), p(sh.gphi), 1, p(sh.S), p(sh.rhs), None))
        n = sh.n_red
        return torch.tril(sh.S[:n, :n]).cpu().numpy(), sh.rhs[:n].cpu().numpy(), sh.Xphi.cpu().numpy().copy()
    S_tile, rhs_tile, xphi_tile = build()
    os.environ['BA_SCHUR_POINTS'] = '1'
    try:
        S_pts, rhs_pts, xphi_pts = build()
    finally:
        os.environ.pop('BA_SCHUR_POINTS', None)
    assert np.abs(S_tile - S_pts).max() <= 1e-11 * np.abs(S_pts).max()
    assert np.abs(rhs_tile - rhs_pts).max() <= 1e-11 * np.abs(rhs_pts).max()
    assert np.abs(xphi_tile - xphi_pts).max() <= 1e-11 * np.abs(xphi_pts).max()
    # the point kernel with room for 5 observations per wave: longer tracks go through in chunks (5 + 5 + 4)
    os.environ['BA_SCHUR_POINTS'], os.environ['BA_SCHUR_POINT_ROWS'] = '1', '5'
    try:
        S_chk, rhs_chk, xphi_chk = build()
    finally:
        os.environ.pop('BA_SCHUR_POINTS', None), os.environ.pop('BA_SCHUR_POINT_ROWS', None)
    assert np.abs(S_chk - S_pts).max() <= 1e-11 * np.abs(S_pts).max()
    assert np.abs(rhs_chk - rhs_pts).max() <= 1e-11 * np.abs(rhs_pts).max()
    assert np.abs(xphi_chk - xphi_pts).max() <= 1e-11 * np.abs(xphi_pts).max()
    # the reduced system against the dense normal equations (points nobody sees have no unknowns there)
    seen = deg > 0
    r, J = orc.dense_jacobian(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'], sc['W'], sc['H'], i, j, uv)
    H, g = J.T @ J, J.T @ r
    H = H + lam * np.diag(np.diag(H))
    pts = np.repeat(seen, 3)
    Hpp, Hpc, Hcc = H[:3 * N][:, :3 * N][np.ix_(pts, pts)], H[:3 * N][pts][:, 3 * N:], H[3 * N:, 3 * N:]
    S_ref = Hcc - Hpc.T @ np.linalg.solve(Hpp, Hpc)
    rhs_ref = -(g[3 * N:] - Hpc.T @ np.linalg.solve(Hpp, g[:3 * N][pts]))
    # the library orders the camera unknowns [C_j | Theta_j] per camera, the dense system [all C | all Theta | phi]
    perm = np.r_[[3 * jj + k if k < 3 else 3 * K + 3 * jj + k - 3 for jj in range(K) for k in range(6)], 6 * K]
    S_ref, rhs_ref = S_ref[np.ix_(perm, perm)], rhs_ref[perm]
    assert np.abs(S_tile - np.tril(S_ref)).max() <= 1e-9 * np.abs(S_ref).max()
    assert np.abs(rhs_tile - rhs_ref).max() <= 1e-9 * np.abs(rhs_ref).max()
    sh.close()


@pytest.mark.gpu
@pytest.mark.parametrize('N,K,track,tile_obs,relabel', [(8000, 40, 10, None, 1), (8000, 40, 4, None, 1), (6800, 80, 10, None, 1),
                                                        (3000, 300, 10, '512', 1), (20000, 100, 10, None, 1),
                                                        (20000, 100, 10, None, 7), (8000, 41, 10, None, 9)])
def test_schur_gram_kernel_equals_pair_kernel(ba, N, K, track, tile_obs, relabel):
    """Tiles of up to 15 cameras go through the Gram-matrix kernel on the matrix cores, the others through the pair
    kernel beside it (a closed orbit has both kinds: the tiles at its seam hold the first and the last cameras).
    Same S, right-hand side, point x phi column and block profile as with the pair kernel alone; some points lose
    observations at random, some all of them."""
    import os
    sc = ba.scenes.orbit_scene(N=N, K=K, track_len=track, seed=5)
    rng = np.random.default_rng(N + K)
    keep = rng.random(sc['i'].size) < rng.choice([1.0, 0.7, 0.0], size=N, p=[0.7, 0.27, 0.03])[sc['i']]
    i, j, uv = sc['i'][keep], sc['j'][keep], sc['uv'][keep]
    C_0, Theta_0 = sc['C_0'], sc['Theta_0']
    if relabel > 1:     # camera numbers scattered (a tile's cameras are then a list, not a range of numbers)
        perm = (np.arange(K) * relabel) % K
        assert np.unique(perm).size == K
        j = perm[j].astype(j.dtype)
        order = np.lexsort((j, i))
        i, j, uv = i[order], j[order], uv[order]
        C_0, Theta_0 = np.empty_like(C_0), np.empty_like(Theta_0)
        C_0[perm], Theta_0[perm] = sc['C_0'], sc['Theta_0']
    if tile_obs:
        os.environ['BA_TILE_OBS'] = tile_obs
    try:
        sh = ba.engine.Shard(i, j, uv, N, K, sc['W'], sc['H'], DEV)
    finally:
        os.environ.pop('BA_TILE_OBS', None)
    sh.set_params(sc['X_0'], C_0, Theta_0, sc['phi_x_0'])
    sh.eval_jtj()
    lib, p = sh.lib, ba.engine.ptr
    first = torch.zeros(sh.ld_red // 64 + 1, dtype=torch.int32, device=DEV)

    def build():
        sh.S.zero_()
        ba.engine.check(lib.ba_schur_build_tracked(sh.handle, p(sh.X), 1e-3, p(sh.V), p(sh.Xphi), p(sh.gX), p(sh.B),
                                                   p(sh.bphi), p(sh.hphi), p(sh.gC), p(sh.gTheta), p(sh.gphi), 1, p(sh.S),
                                                   p(sh.rhs), p(first), 1, ba.engine._lib.stream_ptr()))
        n = sh.n_red
        return (torch.tril(sh.S[:n, :n]).cpu().numpy(), sh.rhs[:n].cpu().numpy(), sh.Xphi.cpu().numpy().copy(),
                first.cpu().numpy().copy())
    got = build()
    os.environ['BA_SCHUR_NO_MFMA'] = '1'
    try:
        want = build()
    finally:
        os.environ.pop('BA_SCHUR_NO_MFMA', None)
    for a, b in zip(got[:3], want[:3]):
        assert np.isfinite(a).all()
        assert np.abs(a - b).max() <= 1e-11 * np.abs(b).max()
    assert (got[3] == want[3]).all()
    sh.close()


def _numpy_lm(sc, iters, lam):
    """The same LM loop with dense NumPy normal equations (reference for the trajectory)."""
    X, C, Th = orc.align_scene(sc['X_0'], sc['C_0'], Theta=sc['Theta_0'])
    X, C = orc.normalize_scene(X, C)
    phi = sc['phi_x_0']
    N, K = X.shape[0], C.shape[0]
    args = (sc['W'], sc['H'], sc['i'], sc['j'], sc['uv'])
    r, J = orc.dense_jacobian(X, C, Th, phi, *args)
    E = [r @ r / (2 * r.size / 2)]
    for _ in range(iters):
        H, g = J.T @ J, J.T @ r
        d = np.linalg.solve(H + lam * np.diag(np.diag(H)), -g)
        Xn = X + d[:3 * N].reshape(N, 3)
        Cn = C + d[3 * N:3 * N + 3 * K].reshape(K, 3)
        Tn = Th + d[3 * N + 3 * K:3 * N + 6 * K].reshape(K, 3)
        pn = phi + d[-1]
        Xn, Cn, Tn = orc.align_scene(Xn, Cn, Theta=Tn)
        Xn, Cn = orc.normalize_scene(Xn, Cn)
        rn, Jn = orc.dense_jacobian(Xn, Cn, Tn, pn, *args)
        En = rn @ rn / (2 * rn.size / 2)
        if En < E[-1]:
            X, C, Th, phi, r, J = Xn, Cn, Tn, pn, rn, Jn
            E.append(En)
            lam = max(lam / 3.0, 1e-12)
        else:
            E.append(E[-1])
            lam *= 4.0
    return X, C, Th, phi, np.array(E)


def test_fit_scene_lm_converges_like_dense_lm(ba):
    sc = ba.scenes.orbit_scene(N=300, K=80, track_len=8, seed=4)
    assert sc['i'].size > 1500
    out = ba.fit_scene_sparse(sc['i'], sc['j'], sc['uv'], sc['N_local'], sc['K'], sc['W'], sc['H'], 1e-3, 9,
                              sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'], optimizer='LM', print_step=10 ** 9)
    X, C, Th, phi, E = out
    Xr, Cr, Tr, pr, Er = _numpy_lm(sc, 8, 1e-3)
    assert E.shape == Er.shape
    assert rel_err(E, Er) < 1e-6
    assert E[-1] < 0.2 and E[-1] < 1e-2 * E[0]          # from the noisy start down to the pixel-quantisation floor
    assert np.abs(X - Xr).max() < 1e-6 and np.abs(C - Cr).max() < 1e-6
    assert np.abs(Th - Tr).max() < 1e-6 and abs(phi - pr) < 1e-6
    with pytest.raises(ValueError, match="optimizer='LM' needs"):
        ba.fit_scene_sparse(sc['i'], sc['j'], sc['uv'], sc['N_local'], sc['K'], sc['W'], sc['H'], 1e-3, 5, sc['X_0'],
                            sc['C_0'], sc['Theta_0'], sc['phi_x_0'], optimizer='LM', phi_x_mask=False)


def test_fit_scene_lm_on_the_recorded_scene_with_long_tracks(ba, golden):
    """moscow_ride (93 points, 327 cameras, tracks of up to 196 observations: the chunked point kernel of the
    Schur build) from the recorded start of the reference: the accept/reject trajectory of the dense NumPy loop,
    and an energy below what the reference's Adam reached in 300 iterations."""
    g = golden('f2_moscow_ride.npz')
    sc = dict(i=g['obs_i'], j=g['obs_j'], uv=g['obs_xy'] + 0.5, W=int(g['W']), H=int(g['H']), X_0=g['free__X_0'],
              C_0=g['C_0'], Theta_0=g['Theta_0'], phi_x_0=float(g['phi_x_0']))
    assert np.bincount(sc['i']).max() > 128
    iters = 12
    out = ba.fit_scene_sparse(sc['i'], sc['j'], sc['uv'], int(g['N']), int(g['K']), sc['W'], sc['H'], 1e-3, iters,
                              sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'], optimizer='LM', print_step=10 ** 9)
    Xr, Cr, Tr, pr, Er = _numpy_lm(sc, iters - 1, 1e-3)
    assert out[4].shape == Er.shape
    assert rel_err(out[4], Er) < 1e-6
    assert out[4][-1] < g['free__E'][-1] < out[4][0]
    assert np.abs(out[0] - Xr).max() < 1e-5 * np.abs(Xr).max() and np.abs(out[1] - Cr).max() < 1e-5


@pytest.mark.parametrize('track_len,N', [(10, 60000), (25, 30000)], ids=['band-of-one-block', 'band-of-three-blocks'])
def test_lm_loop_at_two_thousand_cameras_is_the_same_with_and_without_chains(ba, track_len, N):
    """n = 12 001 unknowns (2 000 cameras): the reduced system is cut into two nested levels of chains, factorised
    with look-ahead on side streams and the fused diagonal-block kernel, solved backwards in a few launches, without a
    host wait in between.  The same LM run with all of that switched off (natural order, one stream, a launch per
    block step: the path the NumPy comparisons of the other tests pin) must give the same energies and parameters."""
    import ctypes as C
    K = 2000
    sc = ba.scenes.orbit_scene(N=N, K=K, track_len=track_len, seed=5)      # (tracks of 25 frames: a wider band, the
                                                                            # general look-ahead instead of the fused kernel)

    def run(env):
        os.environ.update(env)
        try:
            sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], K, sc['W'], sc['H'], DEV)
            sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
            fit = ba.lm.LMFit(sh, 7, lam=1e-3, print_step=10 ** 9, printer=lambda *a: None).run()
            plan = _last_plan(sh.lib)
            out = (fit.energies(), sh.points_to_caller(sh.X).cpu().numpy(), sh.cam.cpu().numpy(), plan)
            sh.close()
            return out
        finally:
            for k in env:
                os.environ.pop(k, None)

    E, X, cam, plan = run({})
    E0, X0, cam0, plan0 = run({'BA_CHOL_NO_CHAINS': '1', 'BA_CHOL_BACK_STEPS': '1', 'BA_SCHUR_FOUR_POINTS': '1'})
    assert len(plan) >= (2 if track_len == 10 else 1) and plan[0][1] > 4 and len(plan0) == 0
    assert (plan[0][3] == 1) == (track_len == 10)
    assert E.shape == E0.shape and rel_err(E, E0) < 1e-9
    assert E[-1] < 1e-2 * E[0]
    assert np.abs(X - X0).max() < 1e-8 * np.abs(X0).max() and np.abs(cam - cam0).max() < 1e-8


def test_lm_loop_recovers_when_the_assumed_profile_is_too_narrow(ba):
    """The LM loop queues its solves on the profile of the solve before.  A scene with the same number of cameras
    but longer tracks (a wider band) right after a scene with short ones: the device-side guard voids the first
    step, the loop clears S, reads the profile again and goes on - same energies as a second run of that scene."""
    K = 260
    narrow = ba.scenes.orbit_scene(N=1500, K=K, track_len=4, seed=2)
    wide = ba.scenes.orbit_scene(N=1500, K=K, track_len=30, seed=3)

    def run(sc, iters=5):
        sh = ba.engine.Shard(sc['i'], sc['j'], sc['uv'], sc['N_local'], K, sc['W'], sc['H'], DEV)
        sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
        voided = []
        pivot = sh.lm_pivot
        sh.lm_pivot = lambda: (voided.append(pivot()), voided[-1])[1]
        fit = ba.lm.LMFit(sh, iters + 1, lam=1e-3, print_step=10 ** 9, printer=lambda *a: None).run()
        E = fit.energies()
        sh.close()
        return E, voided

    E_n, v_n = run(narrow)
    E_w1, v_w1 = run(wide)
    E_w2, v_w2 = run(wide)
    assert -1 in v_w1 and v_w1.count(-1) == 1          # noticed once, then the profile was read again
    assert -1 not in v_w2 and -1 not in v_n
    assert E_w1.shape == E_w2.shape and rel_err(E_w1, E_w2) < 1e-9
    assert E_w1[-1] < 1e-2 * E_w1[0]


def test_twenty_thousand_cameras_lm_step_at_full_size(ba):
    """The large-K configuration at its full size (K = 20 000 cameras, 1 M points, 10 M observations: reduced
    system of 120 001 unknowns, S = 115 GB in HBM) through size-independent properties of ONE damped
    Gauss-Newton step: the factorisation succeeds; S d = rhs holds on block rows of S saved before the in-place
    factorisation (first, middle and last rows: element offsets beyond 2^32); six rows of S of a late camera equal
    the Schur complement formed from per-observation Jacobians of the CPU restatement; the back-substituted
    point steps satisfy the point rows of the normal equations; the energy decreases."""
    K, N, lam = 20000, 1_000_000, 1e-3
    sc = ba.scenes.orbit_scene(N=N, K=K, track_len=10, seed=0)
    i, j, uv, W, H = sc['i'], sc['j'], sc['uv'], sc['W'], sc['H']
    sh = ba.engine.Shard(i, j, uv, N, K, W, H, DEV)
    sh.set_params(sc['X_0'], sc['C_0'], sc['Theta_0'], sc['phi_x_0'])
    sh.gauge_fix()
    sh.eval_jtj()
    E0 = sh.energy()
    n, ld = sh.n_red, sh.ld_red
    assert n == 6 * K + 1 and ld * ld * 8 > 100e9
    j_star = K - 7                                       # a late camera: rows 6 j* .. 6 j* + 5
    starts = [0, (3 * K // 64) * 64, (6 * j_star // 64) * 64, ld - 64]
    saved = {}

    def keep(S, rhs):
        saved['rhs'] = rhs.clone()
        for r0 in starts:
            saved[r0] = (S[r0:r0 + 64, :].clone(), S[:, r0:r0 + 64].clone())

    assert sh.lm_step(lam, on_system=keep) == 0
    d = sh.rhs.clone()                                    # the solution, reduced-system order
    assert bool(torch.isfinite(d).all())
    worst = 0.0
    for r0 in starts:
        rows, cols = saved[r0]
        for k in range(64):
            r = r0 + k
            lhs = torch.dot(rows[k, :r + 1], d[:r + 1]) + torch.dot(cols[r + 1:, k], d[r + 1:])
            scale = torch.dot(rows[k, :r + 1].abs(), d[:r + 1].abs()) + torch.dot(cols[r + 1:, k].abs(), d[r + 1:].abs())
            worst = max(worst, float((lhs - saved['rhs'][r]).abs() / (scale + saved['rhs'][r].abs() + 1e-300)))
    assert worst < 1e-9, worst

    # current parameters in the caller's order, and the per-observation Jacobians of the restatement
    X = sh.points_to_caller(sh.X).cpu().numpy()
    C, Th, phi = sh.C.cpu().numpy(), sh.Theta.cpu().numpy(), float(sh.phi.item())
    factors = orc.camera_rotation_tables(Theta=Th)
    R = factors[1] @ factors[0] @ factors[2]
    f = W / 2.0 / np.tan(phi / 2.0)

    def blocks(sel):
        r, J_X, J_T, J_p = orc.jacobian_blocks(X, C, R, *factors, f, phi, W, H, i[sel], j[sel], uv[sel])
        return r, J_X, np.concatenate([-J_X, J_T], axis=2), J_p          # J_cam = [dr/dC | dr/dTheta] (2 x 6)

    # (a) rows of S of camera j*: D' - sum_i W_ij*^T V'_i^-1 W_ij' over the cameras j' <= j*
    pts = np.unique(i[j == j_star])
    sel = np.isin(i, pts)
    r_o, J_X, J_c, J_p = blocks(sel)
    io, jo = i[sel], j[sel]
    want = np.zeros((6, n))
    mine = jo == j_star
    B = np.einsum('vab,vac->bc', J_c[mine], J_c[mine])
    want[:, 6 * j_star:6 * j_star + 6] = B + lam * np.diag(np.diag(B))
    for pt in pts:
        o = np.nonzero(io == pt)[0]
        V = np.einsum('vab,vac->bc', J_X[o], J_X[o])
        Vi = np.linalg.inv(V + lam * np.diag(np.diag(V)))
        a = o[jo[o] == j_star][0]
        Wa = J_X[a].T @ J_c[a]                                               # 3 x 6
        for b in o:
            want[:, 6 * jo[b]:6 * jo[b] + 6] -= Wa.T @ Vi @ (J_X[b].T @ J_c[b])
    r0 = (6 * j_star // 64) * 64
    got = saved[r0][0][6 * j_star - r0:6 * j_star - r0 + 6, :n].cpu().numpy()
    lower = np.arange(n)[None, :] <= (6 * j_star + np.arange(6))[:, None]    # the valid (lower) triangle
    assert np.abs(np.where(lower, got - want, 0.0)).max() < 1e-9 * np.abs(want).max()
    assert np.count_nonzero(np.where(lower, want, 0.0)) > 36                # other cameras' blocks were compared

    # (b) point rows of the damped normal equations for a sample of points
    dX = sh.points_to_caller(sh.dX).cpu().numpy()
    dcam = sh.dcam.cpu().numpy()
    dC, dT, dphi = dcam[:3 * K].reshape(K, 3), dcam[3 * K:6 * K].reshape(K, 3), dcam[6 * K]
    sample = np.random.default_rng(0).choice(N, 300, replace=False)
    sel = np.isin(i, sample)
    r_o, J_X, J_c, J_p = blocks(sel)
    io, jo = i[sel], j[sel]
    dcam_o = np.concatenate([dC[jo], dT[jo]], axis=1)
    lin = np.einsum('vab,vb->va', J_c, dcam_o) + J_p * dphi + r_o           # residual change the camera side causes
    worst = 0.0
    for pt in sample:
        o = io == pt
        if not o.any():
            continue
        V = np.einsum('vab,vac->bc', J_X[o], J_X[o])
        lhs = (V + lam * np.diag(np.diag(V))) @ dX[pt] + np.einsum('vab,va->b', J_X[o], lin[o])
        worst = max(worst, np.abs(lhs).max() / (np.abs(V).max() * np.abs(dX[pt]).max() + 1e-300))
    assert worst < 1e-8, worst

    # (c) the step lowers the energy
    sh.X.add_(sh.dX)
    sh.cam.add_(sh.dcam)
    sh.invalidate_table()
    sh.gauge_fix()
    sh.eval()
    E1 = sh.energy()
    assert E1 < 0.5 * E0, (E0, E1)
    sh.close()
