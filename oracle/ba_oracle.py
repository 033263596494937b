"""CPU oracle for the bundle-adjustment hot path of `modules/pix2world.py` (`p2w`).

TEST INFRASTRUCTURE ONLY.  This module is the checker the HIP path is compared
against.  It may be imported from `tests/`, from `__graft_entry__.smoke()` and
from the `cpu_baseline` leg of `bench.py`, and from nowhere else: the product
package never imports it and has no CPU fallback.

It restates the reference's algorithm on the visibility set only (O(|V|) work and
memory instead of the reference's dense O(N*K)), in float64 NumPy, following the
reference's own formulas line by line (including the way it rebuilds Rx/Ry/Rz
from the entries of R after the first iteration).  Parity is PINNED: the golden
vectors under `tests/golden/` were produced by importing the reference itself
(`oracle/capture_golden.py`) and `tests/test_oracle_golden.py` checks this file
against every one of them.

Every function cites the reference lines it follows.
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "visibility_coo", "rx", "ry", "rz", "ryxz", "get_rotations", "align_scene",
    "normalize_scene", "sgd", "gd_adam", "GD_FUNCS", "lr_manager", "form_params_dict",
    "forward_obs", "energy_and_gradients", "fit_scene", "transform_obs", "transform_dense",
    "error", "count_components", "dense_jacobian", "camera_rotation_tables",
]


# --------------------------------------------------------------------------- #
# visibility set  (p2w:506-508, 531)
# --------------------------------------------------------------------------- #
def visibility_coo(X_pix):
    """COO form of `X_visible = X_pix[:,:,0] >= 0` (p2w:506) in the C order NumPy's boolean
    mask indexing walks (i-major, j ascending), and the pixel centres
    `X_pix[X_visible] + 0.5` (p2w:531).  Returns (i int32[V], j int32[V], uv float64[V,2])."""
    X_pix = np.asarray(X_pix)
    vis = X_pix[:, :, 0] >= 0
    i, j = np.nonzero(vis)
    uv = X_pix[vis] + 0.5
    return i.astype(np.int32), j.astype(np.int32), np.ascontiguousarray(uv, dtype=np.float64)


def count_components(i, j, N, K):
    """Number of connected components of the bipartite visibility graph on N+K nodes
    (p2w:519-525 builds the dense (N+K)^2 adjacency; only the count matters)."""
    parent = np.arange(N + K)

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    for a, b in zip(np.asarray(i).tolist(), (np.asarray(j) + N).tolist()):
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[max(ra, rb)] = min(ra, rb)
    return len({find(a) for a in range(N + K)})


# --------------------------------------------------------------------------- #
# rotations  (p2w:114-140, 254-267)
# --------------------------------------------------------------------------- #
def rx(theta):
    """p2w:114-120."""
    theta = np.asarray(theta, dtype=np.float64).ravel()
    R = np.zeros((theta.size, 3, 3))
    c, s = np.cos(theta), np.sin(theta)
    R[:, 0, 0] = 1.0
    R[:, 1, 1] = c
    R[:, 2, 2] = c
    R[:, 2, 1] = s
    R[:, 1, 2] = -s
    return R


def ry(theta):
    """p2w:122-128."""
    theta = np.asarray(theta, dtype=np.float64).ravel()
    R = np.zeros((theta.size, 3, 3))
    c, s = np.cos(theta), np.sin(theta)
    R[:, 1, 1] = 1.0
    R[:, 0, 0] = c
    R[:, 2, 2] = c
    R[:, 0, 2] = s
    R[:, 2, 0] = -s
    return R


def rz(theta):
    """p2w:130-136."""
    theta = np.asarray(theta, dtype=np.float64).ravel()
    R = np.zeros((theta.size, 3, 3))
    c, s = np.cos(theta), np.sin(theta)
    R[:, 2, 2] = 1.0
    R[:, 0, 0] = c
    R[:, 1, 1] = c
    R[:, 1, 0] = s
    R[:, 0, 1] = -s
    return R


def ryxz(Theta):
    """R = Ry(theta_y) Rx(theta_x) Rz(theta_z)  (p2w:138-140; Theta columns are x, y, z)."""
    Theta = np.asarray(Theta, dtype=np.float64)
    return ry(Theta[:, 1]) @ rx(Theta[:, 0]) @ rz(Theta[:, 2])


def get_rotations(R):
    """Inverse of `ryxz` by atan2 extraction (p2w:254-267)."""
    C_x = np.sqrt(R[:, 1, 0] ** 2 + R[:, 1, 1] ** 2)
    S_x = -R[:, 1, 2]
    C_y, S_y = R[:, 2, 2] / C_x, R[:, 0, 2] / C_x
    C_z, S_z = R[:, 1, 1] / C_x, R[:, 1, 0] / C_x
    return np.column_stack((np.arctan2(S_x, C_x), np.arctan2(S_y, C_y), np.arctan2(S_z, C_z)))


# --------------------------------------------------------------------------- #
# gauge fix  (p2w:270-333)
# --------------------------------------------------------------------------- #
def align_scene(X, C, Theta=None, R=None, ret_R=False):
    """Move camera 0 to the origin with identity rotation (p2w:270-303)."""
    if Theta is None and R is None:
        raise ValueError('Expected Theta or R to be passed')
    if Theta is not None and R is not None:
        raise ValueError('Expected only Theta or R to be passed, not both')
    X_new = X - C[0]
    C_new = C - C[0]
    if Theta is not None:
        R_zero = ryxz(Theta[0, :].reshape(1, -1))[0].T
    else:
        R_zero = R[0].T
    X_new = X_new @ R_zero.T
    C_new = C_new @ R_zero.T
    R_new = R_zero @ (ryxz(Theta) if Theta is not None else R)
    Theta_new = get_rotations(R_new)
    if not ret_R:
        return X_new, C_new, Theta_new
    return X_new, C_new, Theta_new, R_new


def normalize_scene(X, C, main_indexes=None, dist_scale=1.0, ret_j=False):
    """Scale so that the longest baseline from camera 0 has length `dist_scale` (p2w:306-333)."""
    if main_indexes is None:
        j0 = 0
        j1 = 1 + np.argmax(np.linalg.norm(C[1:] - C[0], axis=1))
    else:
        j0, j1 = main_indexes
    dist = np.linalg.norm(C[j1] - C[j0])
    factor = dist_scale / dist
    out = [X * factor, C * factor]
    if ret_j:
        out.append(j1)
    return out


# --------------------------------------------------------------------------- #
# optimiser core  (p2w:336-477)
# --------------------------------------------------------------------------- #
def sgd(x, grad_x, lr):
    """p2w:336-342."""
    return x - lr * grad_x, {}


def gd_adam(x, grad_x, lr, s, r, t, rho_1=0.9, rho_2=0.999):
    """Adam with bias correction and eps = 1e-12 outside the square root (p2w:344-360)."""
    s_new = rho_1 * s + (1.0 - rho_1) * grad_x
    r_new = rho_2 * r + (1.0 - rho_2) * grad_x ** 2
    s_corr = s_new / (1.0 - rho_1 ** t)
    r_corr = r_new / (1.0 - rho_2 ** t)
    x_new = x - lr * s_corr / (1e-12 + np.sqrt(r_corr))
    return x_new, {'s': s_new, 'r': r_new, 't': t + 1}


GD_FUNCS = {'SGD': sgd, 'Adam': gd_adam}     # p2w:362


def form_params_dict(params, optimizer, ret_all_iters, max_iters):
    """Per-group optimiser state: release snapshot, gd / gd_temp / gd_init (p2w:457-477)."""
    for val in params.values():
        val['release'] = val['main'].copy()
        val['gd'], val['gd_temp'], val['gd_init'] = {}, {}, {}
    if optimizer == 'Adam':
        for val in params.values():
            for slot in ('gd', 'gd_temp', 'gd_init'):
                val[slot]['s'] = np.zeros_like(val['main'])
                val[slot]['r'] = np.zeros_like(val['main'])
                val[slot]['t'] = np.array([1])
    if ret_all_iters:
        for val in params.values():
            val['all_iters'] = max_iters * [None]


def lr_manager(lr, iters, E, E_min, E_min_temp, patience_timer, success_timer,
               began_decreasing, increased, patience, factor, params, ret_all_iters,
               recompute, log=None):
    """LR schedule / snapshot / rollback state machine (p2w:365-454).

    `recompute()` stands for `func(*func_args, **func_kwargs)` (p2w:436); its result is
    returned in a list of length <= 1.  `log`, when given, collects the two event lines
    the reference prints (p2w:405, 430)."""
    results = []
    lr_, iters_, E_min_, E_min_temp_ = lr, iters, E_min, E_min_temp
    pt, st, began, inc = patience_timer, success_timer, began_decreasing, increased
    if iters_ == 0:
        E_min_ = E[iters]
    elif E[iters] < E_min_:
        E_min_ = E[iters]
        began = True
        for val in params.values():
            val['release'] = val['main'].copy()
        pt = patience
        if st > 0:
            st -= 1
        else:
            if log is not None:
                log.append(f'{iters_} : Icrease LR to {lr_ * factor}')
            lr_ *= factor
            st = patience
            inc = True
            E_min_temp_ = E_min_
            for val in params.values():
                for key in val['gd']:
                    val['gd_temp'][key] = val['gd'][key].copy()
                    val['gd'][key] = val['gd_init'][key].copy()
    elif E[iters] > E_min_:
        if inc:
            pt = 0
            iters_ -= 1
        st = patience
        if pt > 0:
            pt -= 1
        else:
            if began:
                pt = patience
            else:
                iters_ -= 1
            if log is not None:
                log.append(f'{iters_} : Decrease LR to {lr_ / factor}')
            lr_ /= factor
            for val in params.values():
                val['main'][:] = val['release']
            results.append(recompute())
            if (not inc) or (inc and E_min_temp_ > E_min_):
                for val in params.values():
                    for key in val['gd']:
                        val['gd'][key] = val['gd_init'][key].copy()
            else:
                for val in params.values():
                    for key in val['gd']:
                        val['gd'][key] = val['gd_temp'][key].copy()
                inc = False
    if ret_all_iters:
        for val in params.values():
            val['all_iters'][iters] = val['main'].copy()
    return (lr_, iters_, E_min_, E_min_temp_, pt, st, began, inc), results


# --------------------------------------------------------------------------- #
# per-observation forward model and gradient  (p2w:544-575, 612, 654-702)
# --------------------------------------------------------------------------- #
def forward_obs(X, C, R, f, W, H, i, j, uv):
    """`inner_transform` restricted to the visible pairs (p2w:555-573):
    d = X_i - C_j, q = R_j^T d, model = f q_xy / q_z + (W/2, H/2), residual = model - uv.
    Returns (residual [V,2], q [V,3], d [V,3])."""
    d = X[i] - C[j]
    q = np.einsum('vl,vlk->vk', d, R[j])
    model = f * q[:, :2] / q[:, 2:3] + np.array([W, H]) / 2.0
    return model - uv, q, d


def camera_rotation_tables(Theta=None, R=None, Theta_for_sign=None):
    """Rx, Ry, Rz factor matrices.  From angles (p2w:557) when `Theta` is given; otherwise
    recovered from the entries of a gauge-fixed R exactly as p2w:665-676 does."""
    if Theta is not None:
        return rx(Theta[:, 0]), ry(Theta[:, 1]), rz(Theta[:, 2])
    K = R.shape[0]
    Rx_, Ry_, Rz_ = rx(np.zeros(K)), ry(np.zeros(K)), rz(np.zeros(K))
    cx = np.sign(np.pi - (Theta_for_sign[:, 0] + np.pi / 2.0) % (2.0 * np.pi)) \
        * np.sqrt(R[:, 1, 0] ** 2 + R[:, 1, 1] ** 2)
    Rx_[:, 1, 1] = Rx_[:, 2, 2] = cx
    Rx_[:, 1, 2] = R[:, 1, 2]
    Rx_[:, 2, 1] = -Rx_[:, 1, 2]
    Ry_[:, 0, 0] = Ry_[:, 2, 2] = R[:, 2, 2] / cx
    Ry_[:, 0, 2] = R[:, 0, 2] / cx
    Ry_[:, 2, 0] = -Ry_[:, 0, 2]
    Rz_[:, 0, 0] = Rz_[:, 1, 1] = R[:, 1, 1] / cx
    Rz_[:, 1, 0] = R[:, 1, 0] / cx
    Rz_[:, 0, 1] = -Rz_[:, 1, 0]
    return Rx_, Ry_, Rz_


def _d_R_d_theta(Rx_, Ry_, Rz_):
    """dR/dtheta_l for l = x, y, z stacked on the last axis (p2w:678-692)."""
    K = Rx_.shape[0]
    D_Rx, D_Ry, D_Rz = np.zeros((3, K, 3, 3))
    D_Rx[:, 1, 1] = D_Rx[:, 2, 2] = Rx_[:, 1, 2]
    D_Rx[:, 2, 1] = Rx_[:, 1, 1]
    D_Rx[:, 1, 2] = -D_Rx[:, 2, 1]
    D_Ry[:, 0, 0] = D_Ry[:, 2, 2] = Ry_[:, 2, 0]
    D_Ry[:, 0, 2] = Ry_[:, 0, 0]
    D_Ry[:, 2, 0] = -D_Ry[:, 0, 2]
    D_Rz[:, 0, 0] = D_Rz[:, 1, 1] = Rz_[:, 0, 1]
    D_Rz[:, 1, 0] = Rz_[:, 0, 0]
    D_Rz[:, 0, 1] = -D_Rz[:, 1, 0]
    D = np.zeros((K, 3, 3, 3))
    D[:, :, :, 0] = Ry_ @ D_Rx @ Rz_
    D[:, :, :, 1] = D_Ry @ Rx_ @ Rz_
    D[:, :, :, 2] = Ry_ @ Rx_ @ D_Rz
    return D


def _segment_sum(idx, vals, n):
    out = np.zeros((n, vals.shape[1]))
    for k in range(vals.shape[1]):
        out[:, k] = np.bincount(idx, weights=vals[:, k], minlength=n)
    return out


def jacobian_blocks(X, C, R, Rx_, Ry_, Rz_, f, phi_x, W, H, i, j, uv):
    """Per-observation residual and Jacobian blocks (p2w:654-657, 694-695, 701):
    r [V,2], J_X [V,2,3] (J_C = -J_X), J_Theta [V,2,3], J_phi [V,2]."""
    r, q, d = forward_obs(X, C, R, f, W, H, i, j, uv)
    V = i.size
    A = np.zeros((V, 2, 3))
    A[:, 0, 0] = A[:, 1, 1] = f / q[:, 2]
    A[:, :, 2] = -f / q[:, 2:3] ** 2 * q[:, :2]
    J_X = A @ R[j].transpose(0, 2, 1)
    D_R = _d_R_d_theta(Rx_, Ry_, Rz_)
    D_rot = np.einsum('vnkl,vn->vkl', D_R[j], d)
    J_Theta = A @ D_rot
    J_phi = -f / np.sin(phi_x) * q[:, :2] / q[:, 2:3]
    return r, J_X, J_Theta, J_phi


def energy_and_gradients(X, C, Theta, phi_x, W, H, i, j, uv, R=None, factors=None, f=None,
                         n_visible=None):
    """E (p2w:612) and dE/dX, dE/dC, dE/dTheta, dE/dphi_x (p2w:659-702) on the visible pairs.

    `R` / `factors=(Rx,Ry,Rz)` default to the values built from `Theta` (what the reference
    uses at iteration 0 and after a rollback); pass them to reproduce the later iterations,
    where R comes from `align_scene` and the factors are recovered from its entries."""
    N, K = X.shape[0], C.shape[0]
    phi = float(np.asarray(phi_x).ravel()[0])
    if factors is None:
        factors = camera_rotation_tables(Theta=Theta)
    if R is None:
        R = factors[1] @ factors[0] @ factors[2]
    if f is None:
        f = W / 2.0 / np.tan(phi / 2.0)
    nv = i.size if n_visible is None else n_visible
    r, J_X, J_Theta, J_phi = jacobian_blocks(X, C, R, *factors, f, phi, W, H, i, j, uv)
    E = np.dot(r.ravel(), r.ravel()) / (2 * nv)
    w = r / nv
    gx_obs = np.einsum('vl,vlk->vk', w, J_X)
    gth_obs = np.einsum('vl,vlk->vk', w, J_Theta)
    gX = _segment_sum(i, gx_obs, N)
    gC = -_segment_sum(j, gx_obs, K)
    gTheta = _segment_sum(j, gth_obs, K)
    gphi = np.array([np.dot(w.ravel(), J_phi.ravel())])
    return E, gX, gC, gTheta, gphi


def dense_jacobian(X, C, Theta, phi_x, W, H, i, j, uv):
    """Dense Jacobian of the stacked residual vector (2V rows; columns X (3N), C (3K),
    Theta (3K), phi_x (1)) assembled from the same blocks as the gradient; used to pin the
    J^T J quantities, which have no counterpart in the reference."""
    N, K, V = X.shape[0], C.shape[0], i.size
    phi = float(np.asarray(phi_x).ravel()[0])
    factors = camera_rotation_tables(Theta=Theta)
    R = factors[1] @ factors[0] @ factors[2]
    f = W / 2.0 / np.tan(phi / 2.0)
    r, J_X, J_Theta, J_phi = jacobian_blocks(X, C, R, *factors, f, phi, W, H, i, j, uv)
    J = np.zeros((2 * V, 3 * N + 6 * K + 1))
    rows = np.arange(V)
    for a in range(2):
        for b in range(3):
            J[2 * rows + a, 3 * i + b] = J_X[:, a, b]
            J[2 * rows + a, 3 * N + 3 * j + b] = -J_X[:, a, b]
            J[2 * rows + a, 3 * N + 3 * K + 3 * j + b] = J_Theta[:, a, b]
        J[2 * rows + a, -1] = J_phi[:, a]
    return r.ravel(), J


# --------------------------------------------------------------------------- #
# fit_scene  (p2w:480-726)
# --------------------------------------------------------------------------- #
def fit_scene(i, j, uv, N, K, W, H, lr, max_iters, X_0, C_0, Theta_0, phi_x_0,
              X_mask=True, C_Theta_mask=True, phi_x_mask=True, optimizer='Adam',
              patience=1000, factor=2.0, stop_value=0.0, stop_diff=1e-12,
              ret_all_iters=False, log=None, grad_log=None):
    """`fit_scene` on the COO visibility set: same loop, same update order, same gauge fix,
    same stopping rules and the same return packing as p2w:480-726.  Input checks
    (p2w:501-528) are the caller's business here.  `log` collects the LR-event and
    stop-reason lines; `grad_log` collects (group name, parameters, gradient) per step."""
    nv = i.size
    WH = (W, H)
    X, C, Theta = X_0.copy(), C_0.copy(), Theta_0.copy()
    phi_x = np.array([phi_x_0], dtype=np.float64)
    if X_mask and C_Theta_mask:
        X[:], C[:], Theta[:] = align_scene(X, C, Theta=Theta)
        X[:], C[:] = normalize_scene(X, C)

    state = {}

    def inner_transform(calc_Rs, R=None, f=None):
        # p2w:544-575 on the visible pairs
        out = {}
        if calc_Rs:
            out['Rx'], out['Ry'], out['Rz'] = camera_rotation_tables(Theta=Theta)
        if R is None:
            Rx_, Ry_, Rz_ = (out['Rx'], out['Ry'], out['Rz']) if calc_Rs else state['factors']
            R_ = Ry_ @ Rx_ @ Rz_
            out['R'] = R_
        else:
            R_ = R
        if f is None:
            f_ = W / 2.0 / np.tan(phi_x / 2.0)
            out['f'] = f_
        else:
            f_ = f
        r, q, d = forward_obs(X, C, R_, f_, WH[0], WH[1], i, j, uv)
        return (r, q, d), out

    f = None
    if not phi_x_mask:
        f = W / 2.0 / np.tan(phi_x / 2.0)

    names = []
    if X_mask:
        names.append('X')
    if C_Theta_mask:
        names.extend(['C', 'Theta'])
    if phi_x_mask:
        names.append('phi_x')
    params = {name: {} for name in names}
    if X_mask:
        params['X']['main'] = X
    if C_Theta_mask:
        params['C']['main'], params['Theta']['main'] = C, Theta
    if phi_x_mask:
        params['phi_x']['main'] = phi_x
    form_params_dict(params, optimizer, ret_all_iters, max_iters)

    E = np.zeros(max_iters)
    E_min = E_min_temp = np.inf
    patience_timer, success_timer = 0, patience
    began_decreasing = increased = False
    R = None
    iters = 0
    while iters < max_iters:
        calculate_R = (iters == 0 and X_mask) or not X_mask
        kw = {}
        if not calculate_R:
            kw['R'] = R
        if not phi_x_mask:
            kw['f'] = f
        calculate_Rs = iters == 0 or not X_mask
        (r, q, d), out = inner_transform(calculate_Rs, **kw)
        if calculate_Rs:
            state['factors'] = (out['Rx'], out['Ry'], out['Rz'])
        if calculate_R:
            R = out['R']
        if phi_x_mask:
            f = out['f']

        E[iters] = np.dot(r.ravel(), r.ravel()) / (2 * nv)

        kw = {}
        if not C_Theta_mask:
            kw['R'] = R
        if not phi_x_mask:
            kw['f'] = f
        res, recomputed = lr_manager(lr, iters, E, E_min, E_min_temp, patience_timer,
                                     success_timer, began_decreasing, increased, patience,
                                     factor, params, ret_all_iters,
                                     recompute=lambda: inner_transform(C_Theta_mask, **kw),
                                     log=log)
        (lr, iters, E_min, E_min_temp, patience_timer, success_timer,
         began_decreasing, increased) = res

        if lr == 0.0:
            if log is not None:
                log.append(f'{iters} : Reached lr = 0.0')
            break
        if stop_diff > 0.0 and iters > 2 * patience:
            E_min_diff = E[:iters - 2 * patience].min() - E_min
            if 0.0 < E_min_diff < stop_diff:
                if log is not None:
                    log.append(f'{iters} : Reached stopping difference: {E_min_diff} < {stop_diff}')
                break
        if E[iters] < stop_value:
            if log is not None:
                log.append(f'{iters} : Reached stopping value: {E[iters]} < {stop_value}')
            break
        if iters == max_iters - 1:
            if log is not None:
                log.append(f'{iters} : Reached maximum iterations')
            break

        if recomputed:
            (r, q, d), out = recomputed[0]
            if C_Theta_mask:
                state['factors'] = (out['Rx'], out['Ry'], out['Rz'])
                R = out['R']
            if phi_x_mask:
                f = out['f']

        # p2w:654-702 on the visible pairs
        A = np.zeros((nv, 2, 3))
        A[:, 0, 0] = A[:, 1, 1] = f / q[:, 2]
        A[:, :, 2] = -f / q[:, 2:3] ** 2 * q[:, :2]
        J_X = A @ R[j].transpose(0, 2, 1)
        w = r / nv
        gx_obs = np.einsum('vl,vlk->vk', w, J_X)
        grads = {}
        if X_mask:
            grads['X'] = _segment_sum(i, gx_obs, N)
        if C_Theta_mask:
            if not calculate_Rs and not recomputed:
                state['factors'] = camera_rotation_tables(R=R, Theta_for_sign=Theta)
            D_R = _d_R_d_theta(*state['factors'])
            D_rot = np.einsum('vnkl,vn->vkl', D_R[j], d)
            J_Theta = A @ D_rot
            grads['C'] = -_segment_sum(j, gx_obs, K)
            grads['Theta'] = _segment_sum(j, np.einsum('vl,vlk->vk', w, J_Theta), K)
        if phi_x_mask:
            J_phi = -f / np.sin(phi_x) * q[:, :2] / q[:, 2:3]
            grads['phi_x'] = np.array([np.dot(w.ravel(), J_phi.ravel())])

        for key, val in params.items():
            if grad_log is not None:
                grad_log.append((key, val['main'].copy(), grads[key].copy()))
            new, new_state = GD_FUNCS[optimizer](val['main'], grads[key], lr, **val['gd'])
            val['main'][:] = new
            for k1, v1 in val['gd'].items():
                v1[:] = new_state[k1]

        if X_mask and C_Theta_mask:
            X[:], C[:], Theta[:], R = align_scene(X, C, Theta=Theta, ret_R=True)
            X[:], C[:] = normalize_scene(X, C)
        iters += 1

    if phi_x_mask:
        params['phi_x']['main'] = params['phi_x']['main'][0]
        if ret_all_iters:
            params['phi_x']['all_iters'] = [p[0] for p in params['phi_x']['all_iters']]
    returns = [*[val['main'] for val in params.values()], E[:iters + 1]]
    if ret_all_iters:
        returns.extend([np.array(val['all_iters']) for val in params.values()])
    return returns


# --------------------------------------------------------------------------- #
# scoring helpers  (p2w:192-216, 98-111)
# --------------------------------------------------------------------------- #
def transform_obs(X, C, Theta, phi_x, W, H, i, j, delete_back_points=True,
                  delete_boundary_points=False):
    """`transform` (p2w:192-216) evaluated on the pairs (i, j) only: model pixels [V,2] with
    NaN behind the camera and, optionally, outside the frame."""
    R = ryxz(Theta)
    d = X[i] - C[j]
    q = np.einsum('vl,vlk->vk', d, R[j])
    if delete_back_points:
        q[q[:, 2] <= 0] = np.nan
    with np.errstate(divide='ignore', invalid='ignore'):
        model = (W / np.tan(phi_x / 2.0) * q[:, :2] / q[:, 2:3] + np.array([W, H])) / 2.0
    if delete_boundary_points:
        out = (model[:, 0] < 0) | (model[:, 0] >= W) | (model[:, 1] < 0) | (model[:, 1] >= H)
        model[out] = np.nan
    return model


def transform_dense(X, C, Theta, phi_x, W, H, delete_back_points=True,
                    delete_boundary_points=False):
    """`transform` (p2w:192-216) for all N*K pairs, shape (N, K, 2)."""
    N, K = X.shape[0], C.shape[0]
    ii, jj = np.divmod(np.arange(N * K), K)
    return transform_obs(X, C, Theta, phi_x, W, H, ii, jj, delete_back_points,
                         delete_boundary_points).reshape(N, K, 2)


def error(X_pix, X_model):
    """Energy on the pairs visible in both the data and the model (p2w:98-111)."""
    vis = (X_pix[:, :, 0] >= 0) & ~np.isnan(X_model[:, :, 0])
    diff = X_model[vis].ravel() - (X_pix[vis].ravel() + 0.5)
    return np.dot(diff, diff) / (2 * vis.sum())


# --------------------------------------------------------------------------- #
# linear initialisation steps on either side of the bundle adjustment
# (p2w:1655-1721 get_X_from_C_Theta, p2w:1724-1808 get_C_Theta_from_X), on the visible pairs only
# --------------------------------------------------------------------------- #
def _intrinsics(phi_x, W, H, normalize):
    """f_x = f_y, p_x, p_y and the map of a pixel centre (p2w:1681-1691, 1748-1758)."""
    if not normalize:
        return W / 2.0 / np.tan(phi_x / 2.0), W / 2.0, H / 2.0, 1.0, np.zeros(2)
    m = max(W, H)
    return W / m / 2.0 / np.tan(phi_x / 2.0), 0.0, 0.0, 1.0 / m, 0.5 * np.array([W, H])


def triangulate_obs(i, j, uv, N, C, Theta, phi_x, W, H, normalize=False):
    """`get_X_from_C_Theta` (p2w:1655-1721): linear triangulation of every point seen at least
    twice from the rows u P_3 - P_1, v P_3 - P_2 of its observations (P = K [R^T | -R^T C],
    p2w:1693-1704), solved through the 3x3 normal equations (p2w:1716-1719).  Points whose row
    matrix has rank < 3 by numpy's rule (tolerance: largest singular value x max(2 K0, 4) x eps,
    K0 = cameras that see a point of the selection, p2w:1676, 1707) stay NaN."""
    X = np.full((N, 3), np.nan)
    deg = np.bincount(i, minlength=N)
    sel = deg[i] >= 2                                             # p2w:1671
    if not sel.any():
        return X
    K0 = np.unique(j[sel]).size                                   # p2w:1676-1677
    f, px, py, scale, shift = _intrinsics(phi_x, W, H, normalize)
    c = (uv - shift) * scale                                      # p2w:1682 / 1688
    R = ryxz(Theta)
    RT = R.transpose(0, 2, 1)
    P = np.array([[f, 0, px], [0, f, py], [0, 0, 1.0]]) @ np.concatenate((RT, -(RT @ C[:, :, None])), axis=2)
    rows0 = c[:, 0:1] * P[j, 2, :] - P[j, 0, :]                   # p2w:1703
    rows1 = c[:, 1:2] * P[j, 2, :] - P[j, 1, :]                   # p2w:1704
    tol_rows = max(2 * K0, 4)
    order = np.argsort(i, kind='stable')
    start = np.concatenate(([0], np.cumsum(deg)))
    for pt in np.nonzero(deg >= 2)[0]:
        e = order[start[pt]:start[pt + 1]]
        A = np.concatenate((rows0[e], rows1[e]))
        s = np.linalg.svd(A, compute_uv=False)
        if (s > s.max() * tol_rows * np.finfo(float).eps).sum() < 3:      # matrix_rank, p2w:1707
            continue
        M, b = A[:, :3], A[:, 3]
        X[pt] = -np.linalg.solve(M.T @ M, M.T @ b)                # p2w:1717-1719
    return X


def resect_obs(i, j, uv, K, X, phi_x, W, H, normalize=False, use_gram=None):
    """`get_C_Theta_from_X` (p2w:1724-1808): DLT resection of every camera that sees at least 6
    points.  Rows (-Xh, 0, u Xh), (0, -Xh, v Xh) per observation (p2w:1767-1770), rows of points
    with NaN coordinates zeroed (p2w:1771-1772), rank >= 11 required (p2w:1774), P = right
    singular vector of the smallest singular value (SVD for small problems, smallest eigenpair
    of A^T A for large ones, p2w:1785-1792; use_gram forces one of the two), then K^-1 P, the
    rotation by orthogonalisation, T / mean singular value, C = -R T, Theta from R sign(det R)
    (p2w:1794-1806)."""
    C = np.full((K, 3), np.nan)
    Theta = np.full((K, 3), np.nan)
    deg = np.bincount(j, minlength=K)
    if not (deg >= 6).any():
        raise ValueError('Expected each camera to have at least 6 visible points')      # p2w:1742
    J0 = np.nonzero(deg >= 6)[0]
    N0 = np.unique(i[deg[j] >= 6]).size                           # p2w:1743-1744
    f, px, py, scale, shift = _intrinsics(phi_x, W, H, normalize)
    K_inv = np.array([[1.0 / f, 0.0, -px / f], [0.0, 1.0 / f, -py / f], [0.0, 0.0, 1.0]])
    c = (uv - shift) * scale
    Xh = np.ones((X.shape[0], 4))
    Xh[:, :3] = X
    mats = {}
    for cam in J0:
        e = np.nonzero(j == cam)[0]
        xh = Xh[i[e]]
        A = np.zeros((2 * e.size, 12))
        A[0::2, 0:4] = -xh
        A[1::2, 4:8] = -xh
        A[0::2, 8:12] = c[e, 0:1] * xh
        A[1::2, 8:12] = c[e, 1:2] * xh
        A[np.isnan(A).any(axis=1)] = 0.0
        s = np.linalg.svd(A, compute_uv=False)
        if (s > s.max() * max(2 * N0, 12) * np.finfo(float).eps).sum() >= 11:
            mats[cam] = A
    if not mats:
        return C, Theta
    J1 = np.array(sorted(mats))
    N1 = np.unique(i[np.isin(j, J1)]).size
    gram = (N1 * J1.size > 25000) if use_gram is None else use_gram
    for cam in J1:
        A = mats[cam]
        if gram:
            w, v = np.linalg.eigh(A.T @ A)
            p = v[:, 0]
        else:
            p = np.linalg.svd(A)[2][-1]
        RT_scaled = K_inv @ p.reshape(3, 4)
        U, Sigma, Vh = np.linalg.svd(RT_scaled[:, :3])            # orthogonolize_matrix, p2w:241-242
        Rm = (U @ Vh).T
        T = RT_scaled[:, 3] / Sigma.mean()
        C[cam] = -(Rm @ T)
        Rm = Rm * np.sign(np.linalg.det(Rm))
        Theta[cam] = get_rotations(Rm[None])[0]
    return C, Theta


# --------------------------------------------------------------------------- #
# joint alignment of subscenes  (fit_subscenes_GD, p2w:729-949)
# --------------------------------------------------------------------------- #
def subscene_entries(X):
    """Present entries of X (N, P, 3) (p2w:754: first coordinate not NaN) in C order:
    (i int32[n], j int32[n], x float64[n, 3])."""
    X = np.asarray(X, dtype=np.float64)
    present = ~np.isnan(X[:, :, 0])
    i, j = np.nonzero(present)
    return i.astype(np.int32), j.astype(np.int32), np.ascontiguousarray(X[present])


def subscenes_forward(i, j, x, N, S, T, R):
    """`inner_transform` of fit_subscenes_GD on the present entries (p2w:813-818):
    x_rot = R_j x, x_new = S_j x_rot + T_j, diff = x_new - mean over the point's entries."""
    x_rot = np.einsum('nkl,nl->nk', R[j], x)
    x_new = S[j][:, None] * x_rot + T[j]
    cnt = np.bincount(i, minlength=N).astype(np.float64)
    with np.errstate(invalid='ignore', divide='ignore'):
        mean = _segment_sum(i, x_new, N) / cnt[:, None]
    return x_new - mean[i], x_rot, x_new


def subscenes_energy_and_gradients(i, j, x, N, P, S, T, Psi, R=None, factors=None):
    """Energy (p2w:848-849) and the three gradients as the reference forms them (p2w:886-924;
    note its factor 2/NP against 1/(2 NP) in the energy).  `factors` = (Rx, Ry, Rz)."""
    if factors is None:
        factors = rx(Psi[:, 0]), ry(Psi[:, 1]), rz(Psi[:, 2])
    if R is None:
        R = factors[1] @ factors[0] @ factors[2]
    n = float(len(i))
    diff, x_rot, _ = subscenes_forward(i, j, x, N, S, T, R)
    E = float(np.dot(diff.ravel(), diff.ravel()) / (2.0 * n))
    cnt = np.bincount(i, minlength=N).astype(np.float64)
    with np.errstate(invalid='ignore', divide='ignore'):
        dc = diff - (_segment_sum(i, diff, N) / cnt[:, None])[i]          # p2w:886
    gS = 2.0 / n * np.bincount(j, weights=np.einsum('nk,nk->n', dc, x_rot), minlength=P)      # p2w:888
    gT = 2.0 / n * _segment_sum(j, dc, P)                                                      # p2w:890
    D = _d_R_d_theta(*factors)                                                                 # p2w:905-919
    d_rot = np.einsum('nkml,nm->nkl', D[j], x)                                                 # p2w:921-922
    gPsi = 2.0 / n * _segment_sum(j, np.einsum('nk,nkl->nl', dc, d_rot) * S[j][:, None], P)    # p2w:923-924
    return E, gS, gT, gPsi


def subscenes_gauge(S, T, Psi):
    """Subscene 0 becomes the identity similarity (p2w:932-937).  Returns S, T, Psi, R."""
    R = ryxz(Psi)
    T = (T - T[0]) @ R[0] / S[0]
    R = R[0].T @ R
    Psi = get_rotations(R)
    S = np.abs(S)
    S = S / S[0]
    return S, T, Psi, R


def fit_subscenes(X, lr, max_iters, S_0=None, T_0=None, Psi_0=None, optimizer='Adam', patience=1000,
                  factor=2.0, stop_value=0.0, stop_diff=1e-12, ret_all_iters=False, log=None, trace=None):
    """fit_subscenes_GD (p2w:729-949) on the present entries only.  Returns what the reference
    returns: [X_new (N, P, 3), S, T, Psi, E[:iters+1]] (+ per-iteration copies).  `trace`, when a
    list, receives (iters, S, T, Psi, gS, gT, gPsi) before every update."""
    X = np.asarray(X, dtype=np.float64)
    N, P = X.shape[:2]
    i, j, x = subscene_entries(X)
    S = np.ones(P) if S_0 is None else np.array(S_0, dtype=np.float64).ravel().copy()
    T = np.zeros((P, 3)) if T_0 is None else np.array(T_0, dtype=np.float64).copy()
    Psi = np.zeros((P, 3)) if Psi_0 is None else np.array(Psi_0, dtype=np.float64).copy()
    params = {'S': {'main': S}, 'T': {'main': T}, 'Psi': {'main': Psi}}
    form_params_dict(params, optimizer, ret_all_iters, max_iters)
    n = float(len(i))

    def evaluate(R=None):
        """inner_transform with calc_Rs=True unless R is given (p2w:793-820)."""
        factors = None
        if R is None:
            factors = rx(Psi[:, 0]), ry(Psi[:, 1]), rz(Psi[:, 2])
            R = factors[1] @ factors[0] @ factors[2]
        diff, _, _ = subscenes_forward(i, j, x, N, S, T, R)
        return float(np.dot(diff.ravel(), diff.ravel()) / (2.0 * n)), R, factors

    E = np.zeros(max_iters)
    E_min = E_min_temp = np.inf
    patience_timer, success_timer = 0, patience
    began_decreasing = increased = False
    iters, R = 0, None
    while iters < max_iters:
        first = iters == 0
        E[iters], R_now, factors = evaluate(None if first else R)
        if first:
            R = R_now
        state, redo = lr_manager(lr, iters, E, E_min, E_min_temp, patience_timer, success_timer,
                                 began_decreasing, increased, patience, factor, params, ret_all_iters,
                                 recompute=lambda: evaluate(None), log=log)
        lr, iters, E_min, E_min_temp, patience_timer, success_timer, began_decreasing, increased = state
        if lr == 0.0:
            break
        if stop_diff > 0.0 and iters > 2 * patience:
            d = E[:iters - 2 * patience].min() - E_min
            if 0.0 < d < stop_diff:
                break
        if E[iters] < stop_value:
            break
        if iters == max_iters - 1:
            break
        if redo:
            _, R, factors = redo[0]
        if factors is None:                                             # p2w:892-903
            factors = camera_rotation_tables(R=R, Theta_for_sign=Psi)
        _, gS, gT, gPsi = subscenes_energy_and_gradients(i, j, x, N, P, S, T, Psi, R=R, factors=factors)
        if trace is not None:
            trace.append((iters, S.copy(), T.copy(), Psi.copy(), gS, gT, gPsi))
        for name, g in (('S', gS), ('T', gT), ('Psi', gPsi)):           # p2w:926-930
            val = params[name]
            new, state_new = GD_FUNCS[optimizer](val['main'], g, lr, **val['gd'])
            val['main'][:] = new
            for key, v in state_new.items():
                val['gd'][key][:] = v
        S[:], T[:], Psi[:], R = subscenes_gauge(S, T, Psi)
        iters += 1

    R_fin = ryxz(Psi)
    _, _, x_new = subscenes_forward(i, j, x, N, S, T, R_fin)
    X_new = np.full((N, P, 3), np.nan)
    X_new[i, j] = x_new
    out = [X_new, S, T, Psi, E[:iters + 1]]
    if ret_all_iters:
        out.extend([np.array(params[k]['all_iters']) for k in ('S', 'T', 'Psi')])
    return out


# --------------------------------------------------------------------------- #
# homography misfit of every camera pair  (get_H_diff_matrix, p2w:1945-1995)
# --------------------------------------------------------------------------- #
def h_diff_matrix_obs(i, j, uv, K, normalize=False, W=None, H=None, use_gram=None):
    """For every pair of cameras with at least 10 common points (p2w:1973-1974): homography by DLT
    from the rows (-x1, 0, u2 x1), (0, -x1, v2 x1) (p2w:1976-1979), h = right singular vector of the
    smallest singular value (SVD below 50 points, smallest eigenpair of A^T A through ARPACK from
    there on, p2w:1981-1986; use_gram forces one of the two), and the mean squared transfer error of
    the common points (p2w:1989-1993).  NaN elsewhere.  `uv` are pixel centres (X_pix + 0.5)."""
    from scipy.sparse.linalg import eigsh
    c = np.asarray(uv, dtype=np.float64)
    if normalize:
        c = (c - 0.5 * np.array([W, H])) / max(W, H)              # p2w:1963-1964
    out = np.full((K, K), np.nan)
    pts = [dict() for _ in range(K)]
    for e, (pi, cj) in enumerate(zip(np.asarray(i).tolist(), np.asarray(j).tolist())):
        pts[cj][pi] = e
    for j1 in range(K):
        for j2 in range(j1 + 1, K):
            common = sorted(pts[j1].keys() & pts[j2].keys())
            n = len(common)
            if n < 10:
                continue
            x1 = np.ones((n, 3))
            x1[:, :2] = c[[pts[j1][q] for q in common]]
            x2 = c[[pts[j2][q] for q in common]]
            A = np.zeros((2 * n, 9))
            A[0::2, 0:3] = A[1::2, 3:6] = -x1
            A[0::2, 6:9] = x2[:, 0:1] * x1
            A[1::2, 6:9] = x2[:, 1:2] * x1
            gram = (n >= 50) if use_gram is None else use_gram
            if gram:
                _, h = eigsh(A.T @ A, k=1, which='SM', v0=np.eye(3).ravel())
            else:
                h = np.linalg.svd(A)[2][-1]
            y = x1 @ h.reshape(3, 3).T
            y = y / y[:, [2]]
            d = (x2 - y[:, :2]).ravel()
            out[j1, j2] = out[j2, j1] = np.dot(d, d) / n
    return out
