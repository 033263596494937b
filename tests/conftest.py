"""Shared test plumbing: path set-up, the `gpu` marker, golden-file access."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
        return cache[name]
    return load


@pytest.fixture(scope='session')
def ba():
    """The product package (its directory name starts with a digit, hence importlib)."""
    return importlib.import_module('3d_reconstruction_amd')


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-300)
    return np.abs(a - b).max() / scale


def row_err(a, b, floor=1e-4):
    """Worst per-row relative error of a (rows, components) array: every point's / camera's vector is
    compared with its own magnitude, bounded below by `floor` times the array's largest entry (a row that
    is small because its terms cancel carries the rounding of those terms)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    a, b = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
    top = max(np.abs(b).max(), 1e-300)
    scale = np.maximum(np.abs(b).max(axis=1), floor * top)
    return (np.abs(a - b).max(axis=1) / scale).max()


def logs_match(got, want, rtol=1e-8):
    """Printed lines agree: same text, every number within `rtol` (the reference prints energies
    with full repr precision, so the last digits depend on summation order)."""
    import re
    num = re.compile(r'[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?')
    if len(got) != len(want):
        return False
    for a, b in zip(got, want):
        if num.sub('#', a) != num.sub('#', b):
            return False
        for x, y in zip(num.findall(a), num.findall(b)):
            x, y = float(x), float(y)
            if abs(x - y) > rtol * max(abs(y), 1e-300):
                return False
    return True


def f11_case(golden, which, t):
    """One recorded iterate of golden F11 (the reference's own per-observation Jacobians): the
    observation list, the parameters the Jacobians were taken at, and the Jacobian blocks."""
    g = golden('f11_jacobians.npz')
    if which == 'f1':
        N, K, W, H = int(g['f1_N']), int(g['f1_K']), float(g['f1_W']), float(g['f1_H'])
        i, j, xy = g['f1_obs_i'], g['f1_obs_j'], g['f1_obs_xy']
    else:
        g2 = golden('f2_moscow_ride.npz')
        N, K, W, H = int(g2['N']), int(g2['K']), float(g2['W']), float(g2['H'])
        i, j, xy = g2['obs_i'], g2['obs_j'], g2['obs_xy']
    case = dict(N=N, K=K, W=W, H=H, i=i, j=j, uv=xy.astype(np.float64) + 0.5, it=int(g[which + '_keep'][t]))
    for k in ('step_X', 'step_C', 'step_Theta', 'step_phi_x', 'grad_X', 'grad_C', 'grad_Theta', 'grad_phi_x',
              'J_X', 'J_Theta', 'J_phi', 'r', 'R', 'Rx', 'Ry', 'Rz', 'f'):
        case[k] = g[f'{which}_{k}'][t]
    return case


def normal_equations_from_jacobians(case, lam=None):
    """J^T J quantities assembled from per-observation Jacobian blocks (J_C = -J_X, p2w:697), un-normalised:
    V (N, 6: upper triangle of the point blocks), B (K, 21: upper triangle of the [C | Theta] camera blocks),
    bphi (K, 6), hphi, the gradient sums, and - with `lam` - the damped reduced camera system S, rhs in the
    order [C_j | Theta_j] per camera, then phi_x."""
    import scipy.sparse as sp
    N, K, i, j = case['N'], case['K'], case['i'].astype(np.int64), case['j'].astype(np.int64)
    JX, JT, Jp, r = case['J_X'], case['J_Theta'], case['J_phi'], case['r']
    nV = i.size
    rows, cols, vals = [], [], []
    o = np.arange(nV)
    for a in range(2):
        for b in range(3):
            rows += [2 * o + a] * 3
            cols += [3 * i + b, 3 * N + 6 * j + b, 3 * N + 6 * j + 3 + b]
            vals += [JX[:, a, b], -JX[:, a, b], JT[:, a, b]]
        rows.append(2 * o + a)
        cols.append(np.full(nV, 3 * N + 6 * K))
        vals.append(Jp[:, a])
    n = 3 * N + 6 * K + 1
    J = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(2 * nV, n))
    Hm = (J.T @ J).toarray()
    g = J.T @ r.ravel()
    iu3, iu6 = np.triu_indices(3), np.triu_indices(6)
    out = dict(H=Hm, g=g,
               V=np.array([Hm[3 * p:3 * p + 3, 3 * p:3 * p + 3][iu3] for p in range(N)]),
               B=np.array([Hm[3 * N + 6 * c:3 * N + 6 * c + 6, 3 * N + 6 * c:3 * N + 6 * c + 6][iu6] for c in range(K)]),
               bphi=np.array([Hm[3 * N + 6 * c:3 * N + 6 * c + 6, -1] for c in range(K)]), hphi=Hm[-1, -1],
               Xphi=Hm[:3 * N, -1].reshape(N, 3))
    if lam is not None:
        Hd = Hm + lam * np.diag(np.diag(Hm))
        seen = np.repeat(np.bincount(i, minlength=N) > 0, 3)
        Hpp, Hpc, Hcc = Hd[:3 * N, :3 * N][np.ix_(seen, seen)], Hd[:3 * N][seen][:, 3 * N:], Hd[3 * N:, 3 * N:]
        out['S'] = Hcc - Hpc.T @ np.linalg.solve(Hpp, Hpc)
        out['rhs'] = -(g[3 * N:] - Hpc.T @ np.linalg.solve(Hpp, g[:3 * N][seen]))
        out['step'] = np.linalg.solve(Hd[np.ix_(np.r_[seen, np.ones(6 * K + 1, bool)], np.r_[seen, np.ones(6 * K + 1, bool)])],
                                      -g[np.r_[seen, np.ones(6 * K + 1, bool)]])
    return out


def f12_check(g, tag, s, E, gX, gC, gT, gp, tol=1e-9):
    """One teacher-forced evaluation against golden F12 (inputs every other fixture avoids: observations behind
    the camera, pixels at and beyond the int16 limits, a depth of 2^-40, a depth of exactly 0, a camera that sees
    nothing).  Where the reference is finite: energy within `tol`, every gradient row within `tol` of its own
    magnitude.  Where it is not (depth 0): the same rows are non-finite, the others agree as before."""
    want_E = g[f'{tag}__E'][s]
    if np.isfinite(want_E):
        assert abs(E - want_E) <= tol * abs(want_E), (tag, s, E, want_E)
    else:
        assert not np.isfinite(E), (tag, s, E)
    for name, got in (('X', gX), ('C', gC), ('Theta', gT), ('phi_x', gp)):
        key = f'{tag}__grad_{name}'
        if key not in g.files:
            continue
        want = np.asarray(g[key][s], dtype=np.float64).reshape(-1, 3 if name != 'phi_x' else 1)
        got = np.asarray(got, dtype=np.float64).reshape(want.shape)
        bad = ~np.isfinite(want).all(axis=1)
        assert np.array_equal(~np.isfinite(got).all(axis=1), bad), (tag, s, name)
        if (~bad).any():
            w, a = want[~bad], got[~bad]
            scale = np.maximum(np.abs(w).max(axis=1), 1e-300)
            zero = np.abs(w).max(axis=1) == 0.0
            assert (np.abs(a - w).max(axis=1)[~zero] / scale[~zero]).max(initial=0.0) < tol, (tag, s, name)
            assert np.abs(a[zero]).max(initial=0.0) == 0.0, (tag, s, name)
