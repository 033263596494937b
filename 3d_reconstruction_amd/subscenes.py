"""Joint alignment of subscenes by gradient descent: drop-in for the reference's `fit_subscenes_GD`
(p2w:729-949) with the arithmetic on the GPU.

P subscenes hold their own estimates of N points, `X (N, P, 3)` with NaN where a subscene does not
hold a point; the fit finds a similarity per subscene (scale S, shift T, angles Psi) that brings the
copies of every point together, subscene 0 staying the identity.  The loop control is the one of
`fit_scene` (`fit_loop.SceneFit`: learning-rate schedule with snapshot / rollback, stopping rules,
printed lines); only the "shard" behind it differs: `SubsceneSet` keeps the present entries sorted
by point on the device and runs `ba_subscenes_eval` / `ba_subscenes_step`."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check
from .fit_loop import SceneFit, gd_func_dict

F64 = torch.float64


class SubsceneSet:
    """Device state of one alignment problem with the interface `SceneFit` drives:
    eval / wait_energy / step_update / adopt / snapshot / restore.  The parameter vector `cam` is
    [S (P) | T (3P) | Psi (3P)]; there is no point group (`X` is None)."""
    distributed = False
    can_fuse = False

    def __init__(self, X, S_0, T_0, Psi_0, device):
        self.lib = _lib.load()
        self.device = torch.device(device)
        N, P = X.shape[:2]
        self.N, self.P = N, P
        self.K = P                       # size the loop control asks for; the free slice is set by SubsceneFit
        present = ~np.isnan(X[:, :, 0])                                     # p2w:754
        ii, jj = np.nonzero(present)                                        # C order: sorted by point
        self.present = present
        self.n = int(ii.size)
        row_ptr = np.zeros(N + 1, dtype=np.int64)
        np.cumsum(np.bincount(ii, minlength=N), out=row_ptr[1:])
        with torch.cuda.device(self.device):
            self.row_ptr = torch.from_numpy(row_ptr).to(self.device)
            self.ej = torch.from_numpy(jj.astype(np.int32)).to(self.device)
            self.ex = torch.from_numpy(np.ascontiguousarray(X[present], dtype=np.float64)).to(self.device)
            self.cam = torch.from_numpy(np.concatenate([np.ravel(S_0), np.ravel(T_0), np.ravel(Psi_0)])
                                        .astype(np.float64)).to(self.device)
            self.table = torch.empty(40 * P, dtype=F64, device=self.device)
            self.sums = torch.zeros(1 + 7 * P, dtype=F64, device=self.device)
        self.X = None
        self._E_host = C.c_double(0.0)

    @property
    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- what the loop control calls ------------------------------------------------------------
    def snapshot(self):
        return self.cam

    def restore(self, snap):
        self.cam = snap

    def adopt(self, X, cam, X_gauge=None):
        self.cam = cam

    def eval(self, grad_x=False, grad_cam=True, grad_phi=True, timed=False):
        """Energy sum and gradient sums at the current parameters (p2w:843-849, 886-924)."""
        check(self.lib.ba_subscenes_eval(self.N, self.P, self.row_ptr.data_ptr(), self.ej.data_ptr(),
                                         self.ex.data_ptr(), self.cam.data_ptr(), self.table.data_ptr(),
                                         self.sums.data_ptr(), self.stream))

    def wait_energy(self):
        """E = sum of squared differences / (2 n) (p2w:849) of the latest `eval`."""
        check(self.lib.ba_fetch_double(self.sums.data_ptr(), C.byref(self._E_host), self.stream))
        return self._E_host.value / (2.0 * self.n)

    def step_update(self, rule, lr, bc1, bc2, x_free, cam_range, mom_x, mom_cam, gauge, in_place=False):
        """Update rule on S, T, Psi and the gauge (p2w:926-937) into new tensors."""
        cam_new = torch.empty_like(self.cam)
        if mom_cam is not None:
            s_in, r_in = mom_cam
            mom_cam = (torch.empty_like(s_in), torch.empty_like(r_in))
            mp = (s_in.data_ptr(), mom_cam[0].data_ptr(), r_in.data_ptr(), mom_cam[1].data_ptr())
        else:
            mp = (None, None, None, None)
        check(self.lib.ba_subscenes_step(self.P, rule, float(lr), float(bc1), float(bc2), float(self.n),
                                         self.sums.data_ptr(), self.cam.data_ptr(), cam_new.data_ptr(), *mp,
                                         self.stream))
        return None, cam_new, mom_x, mom_cam

    # ---- results ----------------------------------------------------------------------------------
    def split(self, v):
        P = self.P
        v = np.asarray(v)
        return v[..., :P].copy(), v[..., P:4 * P].reshape(v.shape[:-1] + (P, 3)).copy(), \
            v[..., 4 * P:].reshape(v.shape[:-1] + (P, 3)).copy()

    def transformed(self):
        """X_new (N, P, 3) at the current parameters, NaN where the input has no entry (p2w:943-944)."""
        x_new = torch.empty_like(self.ex)
        check(self.lib.ba_subscenes_transform(self.P, self.n, self.ej.data_ptr(), self.ex.data_ptr(),
                                              self.cam.data_ptr(), self.table.data_ptr(), x_new.data_ptr(),
                                              self.stream))
        out = np.full((self.N, self.P, 3), np.nan)
        out[self.present] = x_new.cpu().numpy()
        return out


class SubsceneFit(SceneFit):
    """`SceneFit` on a `SubsceneSet`: one free group (all 7P parameters), the gauge inside the step."""

    def __init__(self, subscenes, lr, max_iters, **kw):
        super().__init__(subscenes, lr, max_iters, X_mask=False, C_Theta_mask=True, phi_x_mask=True, fuse=False, **kw)
        self.cam_range = (0, 7 * subscenes.P)

    # a callable from the optimiser registry sees the groups S, T, Psi in that order (p2w:926-930)
    def _host_group_shapes(self):
        P = self.sh.P
        return [('S', (P,)), ('T', (P, 3)), ('Psi', (P, 3))]

    def _host_fetch(self):
        sh, P = self.sh, self.sh.P
        cam = sh.cam.cpu().numpy().copy()
        g = sh.sums.cpu().numpy()[1:1 + 7 * P] * (2.0 / float(sh.n))        # p2w:888, 890, 923
        cut = lambda v: {'S': v[:P], 'T': v[P:4 * P].reshape(P, 3), 'Psi': v[4 * P:].reshape(P, 3)}
        return cam, cut(cam), cut(g)

    def _host_store(self, cam, main):
        sh = self.sh
        sh.cam = torch.from_numpy(cam).to(sh.device)
        # the gauge of p2w:932-937 is the tail of the step kernel: a plain step of length zero leaves only that
        _, sh.cam, _, _ = sh.step_update(1, 0.0, 1.0, 1.0, False, self.cam_range, None, None, True)


def fit_subscenes_GD(X, lr, max_iters, S_0=None, T_0=None, Psi_0=None, optimizer='Adam', patience=1000,
                     factor=2.0, stop_value=0.0, stop_diff=1e-12, print_step=1000, ret_all_iters=False, device=None):
    """Drop-in for the reference's `fit_subscenes_GD` (p2w:729-949): same arguments, checks and
    messages, same printed lines, returns [X_new, S, T, Psi, E[:iters+1]] (+ the per-iteration
    copies of S, T, Psi with `ret_all_iters`)."""
    X = np.asarray(X)
    if X.ndim != 3:                                                         # p2w:744-752
        raise ValueError(f'Expected X to have ndim = 3, got {X.ndim}')
    if X.shape[2] != 3:
        raise ValueError(f'Expected X to have shape[2] = 3, got {X.shape[2]}')
    N, P = X.shape[:2]
    if N < 3:
        raise ValueError(f'Expected X to have shape[0] >= 3, got {N}')
    if P < 2:
        raise ValueError(f'Expected X to have shape[1] >= 2, got {P}')
    present = ~np.isnan(X[:, :, 0])                                         # p2w:754-759
    if 3 * present.sum() < 7 * (P - 1):
        raise ValueError('Unable to solve the problem, the number of equations is smaller than the number of degrees of freedom')
    # every subscene shares at least 3 points with some other subscene (p2w:761-765)
    common = present.T.astype(np.int64) @ present.astype(np.int64)
    np.fill_diagonal(common, 0)
    loose = np.where((common >= 3).sum(axis=0) == 0)[0]
    if loose.size > 0:
        raise ValueError(f'Unable to solve the problem, subscenes {loose} are not properly connected to the other subscenes')
    if optimizer not in gd_func_dict:                                       # p2w:767-768
        raise ValueError(f'Unkown name of the optmizer, choose one from {[key for key in gd_func_dict]}')
    if S_0 is None:                                                         # p2w:770-786
        S_0 = np.ones(P)
    elif np.size(S_0) != P:
        raise ValueError(f'Expected S_0 to have size = P, got {np.size(S_0)}')
    if T_0 is None:
        T_0 = np.zeros((P, 3))
    elif np.shape(T_0) != (P, 3):
        raise ValueError(f'Expected T_0 to have shape = (P, 3), got {np.shape(T_0)}')
    if Psi_0 is None:
        Psi_0 = np.zeros((P, 3))
    elif np.shape(Psi_0) != (P, 3):
        raise ValueError(f'Expected Psi_0 to have shape = (P, 3), got {np.shape(Psi_0)}')

    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError('fit_subscenes_GD needs an MI355X: no GPU is visible and there is no CPU fallback')
        device = 'cuda:0'
    sub = SubsceneSet(np.asarray(X, dtype=np.float64), S_0, T_0, Psi_0, device)
    fit = SubsceneFit(sub, lr, max_iters, optimizer=optimizer, patience=patience, factor=factor,
                      stop_value=stop_value, stop_diff=stop_diff, print_step=print_step,
                      ret_all_iters=ret_all_iters).run()
    S, T, Psi = sub.split(sub.cam.cpu().numpy())
    out = [sub.transformed(), S, T, Psi, fit.energies()]
    if ret_all_iters:                                                       # p2w:947-948
        hist = fit.history()[0]
        out.extend(list(sub.split(np.array(hist))))
    return out
