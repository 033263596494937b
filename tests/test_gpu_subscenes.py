"""Parity of the device path of `fit_subscenes_GD` (p2w:729-949) with the golden vectors captured from
the reference and with the CPU oracle.  Needs an MI355X: run with `-m gpu`.

Tolerances: energy and gradients within 1e-9 relative at the reference's own iterates, every
iterate of a free run within 1e-9 absolute (the parameters are O(1)), same event log."""
import contextlib
import io

import numpy as np
import pytest
import torch

from conftest import logs_match, rel_err
from oracle import ba_oracle as orc
from test_oracle_golden import F8_RUNS, f8_kwargs

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _sums(ba, sub, S, T, Psi):
    sub.cam = torch.from_numpy(np.concatenate([S.ravel(), T.ravel(), Psi.ravel()])).to(DEV)
    sub.eval()
    torch.cuda.synchronize()
    v = sub.sums.cpu().numpy()
    n, P = sub.n, sub.P
    return v[0] / (2.0 * n), 2.0 / n * v[1:1 + P], 2.0 / n * v[1 + P:1 + 4 * P].reshape(P, 3), \
        2.0 / n * v[1 + 4 * P:].reshape(P, 3)


def test_subscene_gradients_teacher_forced(ba, golden):
    g = golden('f8_subscenes.npz')
    X = g['X']
    P = X.shape[1]
    sub = ba.subscenes.SubsceneSet(X, np.ones(P), np.zeros((P, 3)), np.zeros((P, 3)), DEV)
    for tag in ('adam', 'sgd', 'start'):
        S_, T_, Psi_ = g[f'{tag}_step_S'], g[f'{tag}_step_T'], g[f'{tag}_step_Psi']
        for t in range(0, S_.shape[0], 7):
            E, gS, gT, gPsi = _sums(ba, sub, S_[t], T_[t], Psi_[t])
            assert abs(E - g[f'{tag}_E'][t]) <= 1e-9 * g[f'{tag}_E'][t]
            assert rel_err(gS, g[f'{tag}_grad_S'][t]) < 1e-9
            assert rel_err(gT, g[f'{tag}_grad_T'][t]) < 1e-9
            assert rel_err(gPsi, g[f'{tag}_grad_Psi'][t]) < 1e-9


def test_subscene_step_matches_reference_iterates(ba, golden):
    """One device step from the reference's iterate t lands on its iterate t + 1 (update rule + gauge)."""
    g = golden('f8_subscenes.npz')
    X = g['X']
    P = X.shape[1]
    sub = ba.subscenes.SubsceneSet(X, np.ones(P), np.zeros((P, 3)), np.zeros((P, 3)), DEV)
    S_, T_, Psi_ = g['sgd_step_S'], g['sgd_step_T'], g['sgd_step_Psi']
    for t in (0, 5, 20):
        _sums(ba, sub, S_[t], T_[t], Psi_[t])
        _, cam, _, _ = sub.step_update(1, 1e-2, 1.0, 1.0, False, (0, 7 * P), None, None, False)
        S, T, Psi = sub.split(cam.cpu().numpy())
        assert np.abs(S - S_[t + 1]).max() < 1e-12
        assert np.abs(T - T_[t + 1]).max() < 1e-12
        assert np.abs(Psi - Psi_[t + 1]).max() < 1e-12


@pytest.mark.parametrize('tag', list(F8_RUNS))
def test_fit_subscenes_runs(ba, golden, tag):
    g = golden('f8_subscenes.npz')
    kw = f8_kwargs(g, tag)
    max_iters = kw.pop('max_iters')
    out_txt = io.StringIO()
    with contextlib.redirect_stdout(out_txt):
        out = ba.fit_subscenes_GD(g['X'], kw.pop('lr'), max_iters, ret_all_iters=True, **kw)
    assert out[4].shape == g[f'{tag}_E'].shape
    assert rel_err(out[4], g[f'{tag}_E']) < 1e-9
    for k, name in ((1, 'S'), (2, 'T'), (3, 'Psi')):
        assert np.abs(out[k].reshape(g[f'{tag}_final_{name}'].shape) - g[f'{tag}_final_{name}']).max() < 1e-9
    np.testing.assert_allclose(out[0], g[f'{tag}_X_new'], rtol=0, atol=1e-9, equal_nan=True)
    assert logs_match(out_txt.getvalue().splitlines(), g[f'{tag}_log'].tolist())
    assert out[5].shape == (max_iters, g['X'].shape[1]) and out[7].shape == (max_iters, g['X'].shape[1], 3)


@pytest.mark.parametrize('tag', ['adam', 'sgd', 'events'])
def test_fit_subscenes_with_a_wrapped_registry_entry(ba, golden, tag):
    """The capture hook of the golden files (oracle/capture_golden.py: a recording wrapper in gd_func_dict that
    forwards to the original entry) run against the drop-in: it sees the groups S, T, Psi with the reference's
    parameters and gradients (p2w:926-930), and the run is the reference's."""
    g = golden('f8_subscenes.npz')
    kw = f8_kwargs(g, tag)
    max_iters = kw.pop('max_iters')
    name = kw.get('optimizer', 'Adam')
    orig, seen = ba.gd_func_dict[name], []

    def rec(x, grad, lr_, **state):
        seen.append((np.array(x).copy(), np.array(grad).copy()))
        return orig(x, grad, lr_, **state)
    ba.gd_func_dict[name] = rec
    try:
        with contextlib.redirect_stdout(io.StringIO()) as out_txt:
            out = ba.fit_subscenes_GD(g['X'], kw.pop('lr'), max_iters, **kw)
    finally:
        ba.gd_func_dict[name] = orig
    assert rel_err(out[4], g[f'{tag}_E']) < 1e-9
    assert logs_match(out_txt.getvalue().splitlines(), g[f'{tag}_log'].tolist())
    steps = g[f'{tag}_step_S'].shape[0]
    assert len(seen) == 3 * steps
    for t in range(0, steps, 5):
        for k, n in enumerate(('S', 'T', 'Psi')):
            x, grad = seen[3 * t + k]
            assert np.abs(x - g[f'{tag}_step_{n}'][t]).max() < 1e-9
            assert rel_err(grad, g[f'{tag}_grad_{n}'][t]) < 1e-8
    for k, n in ((1, 'S'), (2, 'T'), (3, 'Psi')):
        assert np.abs(out[k].reshape(g[f'{tag}_final_{n}'].shape) - g[f'{tag}_final_{n}']).max() < 1e-9


def test_fit_subscenes_errors(ba, golden):
    g = golden('f8_subscenes.npz')
    X = g['X']
    with pytest.raises(ValueError, match='ndim = 3'):
        ba.fit_subscenes_GD(X[0], 1e-2, 5)
    with pytest.raises(ValueError, match=r'shape\[1\] >= 2'):
        ba.fit_subscenes_GD(X[:, :1], 1e-2, 5)
    with pytest.raises(ValueError, match='optmizer'):
        ba.fit_subscenes_GD(X, 1e-2, 5, optimizer='LBFGS')
    with pytest.raises(ValueError, match='S_0 to have size'):
        ba.fit_subscenes_GD(X, 1e-2, 5, S_0=np.ones(3))
    Xl = X.copy()
    Xl[2:, 3] = np.nan                      # subscene 3 keeps two points
    with pytest.raises(ValueError, match=r'subscenes \[3\] are not properly connected'):
        ba.fit_subscenes_GD(Xl, 1e-2, 5)


def test_subscenes_larger_than_lds_sums_and_ragged(ba):
    """P large enough for the global-atomics path (7P > 4096), points with one entry or none."""
    rng = np.random.default_rng(3)
    N, P = 3000, 640
    X = np.full((N, P, 3), np.nan)
    for i in range(N):
        if i % 97 == 0:
            continue                         # a point no subscene holds
        js = rng.choice(P, size=1 if i % 50 == 1 else rng.integers(2, 9), replace=False)
        X[i, js] = rng.normal(0, 2.0, (js.size, 3))
    S, T, Psi = np.exp(rng.normal(0, 0.2, P)), rng.normal(0, 1, (P, 3)), rng.normal(0, 0.5, (P, 3))
    sub = ba.subscenes.SubsceneSet(X, S, T, Psi, DEV)
    E, gS, gT, gPsi = _sums(ba, sub, S, T, Psi)
    i, j, x = orc.subscene_entries(X)
    E_o, gS_o, gT_o, gPsi_o = orc.subscenes_energy_and_gradients(i, j, x, N, P, S, T, Psi)
    assert abs(E - E_o) <= 1e-12 * E_o
    assert rel_err(gS, gS_o) < 1e-11 and rel_err(gT, gT_o) < 1e-11 and rel_err(gPsi, gPsi_o) < 1e-11
    _, cam, _, _ = sub.step_update(1, 1e-3, 1.0, 1.0, False, (0, 7 * P), None, None, False)
    S_o, T_o, Psi_o, _ = orc.subscenes_gauge(S - 1e-3 * gS_o, T - 1e-3 * gT_o, Psi - 1e-3 * gPsi_o)
    S_d, T_d, Psi_d = sub.split(cam.cpu().numpy())
    assert np.abs(S_d - S_o).max() < 1e-12 and np.abs(T_d - T_o).max() < 1e-11
    assert np.abs(Psi_d - Psi_o).max() < 1e-12
