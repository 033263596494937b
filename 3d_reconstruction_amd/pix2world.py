"""Drop-in entry points for the bundle-adjustment path of the reference's `modules/pix2world.py`
("p2w"): same names, same arguments, same return packing, same error messages, with the
arithmetic on an MI355X through libba_mi355x.so.

    fit_scene(...)           p2w:480-726   dense (N,K,2) integer tracks, as the reference takes them
    fit_scene_sparse(...)    —             same optimisation on a COO observation list (O(|V|) input)
    fit_scene_sharded(...)   —             the same, every rank of a process group passing only its own points
    transform(...)           p2w:192-216
    error(...)               p2w:98-111
    validate_scene_parameters(...)  p2w:21-95
    sgd, gd_adam             p2w:336-360   the update rules on host arrays (reference arithmetic)
    gd_func_dict             p2w:362       optimiser registry {'SGD': sgd, 'Adam': gd_adam}; the names `optimizer=` accepts

Inputs are never modified; returned arrays are fresh NumPy arrays owned by the caller."""
from __future__ import annotations

import numpy as np
import torch

from . import engine
from .fit_loop import SceneFit, extra_optimizers, gd_adam, gd_func_dict, sgd  # noqa: F401  (registry re-exported)
from .lm import LMFit
from .sharding import all_ranks_ok, checked_partition

__all__ = ['fit_scene', 'fit_scene_sparse', 'fit_scene_sharded', 'transform', 'error', 'validate_scene_parameters', 'gd_func_dict', 'sgd', 'gd_adam',
           'get_X_from_C_Theta', 'get_C_Theta_from_X', 'get_H_diff_matrix', 'ryxz', 'get_rotations', 'align_scene',
           'normalize_scene']


# --------------------------------------------------------------------------------------------- #
# validation (p2w:21-95): same checks in the same order with the same messages
# --------------------------------------------------------------------------------------------- #
def _check_matrix(name, a):
    if a.ndim != 2:
        raise ValueError(f'Expected {name} to have ndim = 2, got {a.ndim}')
    if a.shape[1] != 3:
        raise ValueError(f'Expected {name} to have shape[1] = 3, got {a.shape[1]}')
    if np.isnan(a).any() or np.isinf(a).any():
        raise ValueError(f'Expected {name} not to have nan or inf values')


def validate_scene_parameters(X=None, C=None, Theta=None, R=None, X_pix=None, phi_x=None, ret_shapes=False):
    N = K = None
    if X is not None:
        _check_matrix('X', X)
        if ret_shapes or X_pix is not None:
            N = X.shape[0]
    if C is not None:
        _check_matrix('C', C)
        if ret_shapes or Theta is not None or R is not None or X_pix is not None:
            K = C.shape[0]
    if Theta is not None:
        _check_matrix('Theta', Theta)
        if K is not None and Theta.shape[0] != K:
            raise ValueError(f'Expected Theta to have shape[0] = {K}, got {Theta.shape[0]}')
        elif ret_shapes or R is not None or X_pix is not None:
            K = Theta.shape[0]
    if R is not None:
        if R.ndim != 3:
            raise ValueError(f'Expected R to have ndim = 3, got {R.ndim}')
        if R.shape[1:] != (3, 3):
            raise ValueError(f'Expected R to have shape[1:] = (3, 3), got {R.shape[1:]}')
        if np.isnan(R).any() or np.isinf(R).any():
            raise ValueError('Expected R not to have nan or inf values')
        if (np.abs(R.transpose(0, 2, 1) @ R - np.eye(3)) > 1e-9).any():
            raise ValueError('Expected R[i] @ R[i].T must be approximately equal to eye(3) for all i,'
                             'you may try orthogonolize_matrix(R) to fix it.')
        if (np.abs(np.linalg.det(R) - 1.0) > 1e-9).any():
            raise ValueError('Expected R[i] to have determinant approximately 1.0 for all i,'
                             'you may try R *= np.sign(np.linalg.det(R))[:,np.newaxis,np.newaxis] to fix it.')
        if K is not None and R.shape[0] != K:
            raise ValueError(f'Expected R to have shape[0] = {K}, got {R.shape[0]}')
        elif ret_shapes or X_pix is not None:
            K = R.shape[0]
    if X_pix is not None:
        if X_pix.ndim != 3:
            raise ValueError(f'Expected X_pix to have ndim = 3, got {X_pix.ndim}')
        if X_pix.shape[2] != 2:
            raise ValueError(f'Expected X_pix to have shape[2] = 2, got {X_pix.shape[2]}')
        if N is not None and X_pix.shape[0] != N:
            raise ValueError(f'Expected X_pix to have shape[0] = {N}, got {X_pix.shape[0]}')
        elif ret_shapes:
            N = X_pix.shape[0]
        if K is not None and X_pix.shape[1] != K:
            raise ValueError(f'Expected X_pix to have shape[1] = {K}, got {X_pix.shape[1]}')
        elif ret_shapes:
            K = X_pix.shape[1]
    if phi_x is not None:
        if phi_x <= 0.0 or phi_x >= np.pi:
            raise ValueError(f'Expected phi_x to have a value within (0.0, pi), got {phi_x}')
    if ret_shapes:
        return N, K


def check_solvable(n_visible, N, K, X_mask, C_Theta_mask, phi_x_mask):
    """Mask sanity and the equations-vs-unknowns count (p2w:503-517)."""
    if not X_mask and not C_Theta_mask and not phi_x_mask:
        raise ValueError('At least one of X_mask, C_Theta_mask or phi_x_mask have to be true')
    dof = 0
    if X_mask:
        dof += 3 * N
    if C_Theta_mask:
        dof += 6 * K
    if X_mask and C_Theta_mask:
        dof -= 7
    if phi_x_mask:
        dof += 1
    if 2 * n_visible < dof:
        raise ValueError('Unable to solve the problem, the number of equations is smaller than the number of '
                         'degrees of freedom')


def visibility_graph_components(i, j, N, K, device, group=None):
    """Number of connected components of the bipartite point / camera visibility graph (what the reference asks
    scipy for on a dense (N + K)^2 matrix, p2w:519-523), counted on the device from the COO list.  With several
    ranks every rank passes ITS points (local numbers 0 .. N - 1) and observations; the ranks' graphs meet in the
    cameras, so each rank labels its own graph, the camera labels (K integers per rank) are gathered and merged on
    the host, and the components of points alone are added up."""
    n_cam, n_pts, labels = engine.visibility_components(i, j, N, K, device, want_labels=True)
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return n_cam + n_pts
    world = dist.get_world_size(group)
    gathered = [None] * world
    dist.all_gather_object(gathered, (labels.cpu().numpy(), n_pts), group=group)
    parent = np.arange(K)

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    for lab, _ in gathered:
        for cam in np.nonzero(lab != np.arange(K))[0]:
            a, b = find(int(cam)), find(int(lab[cam]))
            if a != b:
                parent[max(a, b)] = min(a, b)
    return int(sum(1 for cam in range(K) if find(cam) == cam)) + int(sum(n for _, n in gathered))


def check_connected(i, j, N, K, device=None, group=None):
    """The bipartite visibility graph must be one component (p2w:519-525)."""
    n = visibility_graph_components(i, j, N, K, device if device is not None else _default_device(), group)
    if n > 1:
        raise ValueError('Unable to solve the problem, expected the number of connected components of '
                         f'visibility graph to be 1, got {n}')


# --------------------------------------------------------------------------------------------- #
def _default_device():
    if not torch.cuda.is_available():
        raise RuntimeError('3d_reconstruction_amd needs an MI355X visible to PyTorch-ROCm; there is no CPU path')
    return torch.device('cuda', torch.cuda.current_device())


def _pack_results(fit, shard, masks, gather_X=None):
    """Return list as the reference builds it (p2w:718-726)."""
    X_mask, C_Theta_mask, phi_x_mask = masks
    out = []
    if X_mask:
        X = shard.points_to_caller(shard.X)
        X = X if gather_X is None else gather_X(X)
        out.append(X.cpu().numpy().copy())
    if C_Theta_mask:
        out.append(shard.C.cpu().numpy().copy())
        out.append(shard.Theta.cpu().numpy().copy())
    if phi_x_mask:
        out.append(np.float64(shard.phi.item()))
    out.append(fit.energies())
    if fit.ret_all_iters:
        hist = fit.history()
        K3 = 3 * shard.K
        g = 0
        if X_mask:
            if shard.perm is None:
                back = lambda h: h
            else:       # inverse of the engine's point order by a scatter
                inv = np.empty(shard.N, dtype=np.int64)
                inv[shard.perm.cpu().numpy()] = np.arange(shard.N)
                back = lambda h, inv=inv: h[inv]
            snaps = []
            for h in hist[g]:
                if h is not None:
                    h = back(h.reshape(-1, 3))
                    if gather_X is not None:       # several ranks: every snapshot holds this rank's points only
                        t = torch.from_numpy(np.ascontiguousarray(h))
                        dev = getattr(shard, 'device', None)
                        h = gather_X(t if dev is None else t.to(dev)).cpu().numpy()
                snaps.append(h)
            out.append(np.array(snaps))
            g += 1
        if C_Theta_mask:
            out.append(np.array([h[:K3].reshape(-1, 3) for h in hist[g]]))
            out.append(np.array([h[K3:2 * K3].reshape(-1, 3) for h in hist[g]]))
        if phi_x_mask:
            out.append(np.array([h[-1] for h in hist[g]]))
    return out


def fit_scene_sparse(i, j, uv, N, K, W, H, lr, max_iters, X_0, C_0, Theta_0, phi_x_0,
                     X_mask=True, C_Theta_mask=True, phi_x_mask=True, optimizer='Adam',
                     patience=1000, factor=2.0, stop_value=0.0, stop_diff=1e-12, print_step=1000,
                     ret_all_iters=False, device=None, check_connectivity=True, _validate=True):
    """`fit_scene` on a COO observation list: `i`, `j` int arrays (sorted by point, then camera,
    i.e. the order numpy.nonzero gives on the dense visibility mask) and `uv` = pixel centres
    (pixel + 0.5), shape (|V|, 2).  Same optimisation, same returns as `fit_scene`.

    When torch.distributed is initialised with more than one rank, every rank passes the whole
    problem; points (with their observations) are split into contiguous ranges balanced by
    observation count, cameras are replicated, and one all-reduce per evaluation adds the
    camera-side sums.  Every rank returns the full result."""
    X_0, C_0, Theta_0 = (np.asarray(a, dtype=np.float64) for a in (X_0, C_0, Theta_0))
    if _validate:
        n_, k_ = validate_scene_parameters(X=X_0, C=C_0, Theta=Theta_0, phi_x=phi_x_0, ret_shapes=True)
        if n_ != N or k_ != K:
            raise ValueError(f'Expected X_0, C_0 to have shape[0] = {N}, {K}, got {n_}, {k_}')
    dev = torch.device(device) if device is not None else _default_device()
    as_t = lambda a, dt: a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    i_t, j_t, uv_t = as_t(i, None), as_t(j, None), as_t(uv, None)
    nV = int(i_t.numel())
    check_solvable(nV, N, K, X_mask, C_Theta_mask, phi_x_mask)
    if X_mask and C_Theta_mask and check_connectivity:
        # every rank holds the whole list here and labels the whole graph itself (no exchange needed)
        n_cam, n_pts = engine.visibility_components(i_t, j_t, N, K, dev)
        if n_cam + n_pts > 1:
            raise ValueError('Unable to solve the problem, expected the number of connected components of '
                             f'visibility graph to be 1, got {n_cam + n_pts}')
    if optimizer not in gd_func_dict and optimizer not in extra_optimizers:
        raise ValueError(f'Unkown name of the optmizer, choose one from {[key for key in gd_func_dict]}')
    if optimizer == 'LM' and not (X_mask and C_Theta_mask and phi_x_mask):
        raise ValueError("optimizer='LM' needs X_mask, C_Theta_mask and phi_x_mask to be true")

    world = torch.distributed.get_world_size() if (torch.distributed.is_available()
                                                   and torch.distributed.is_initialized()) else 1
    gather_X = None
    if world > 1:
        rank = torch.distributed.get_rank()
        i_host = i_t.cpu().numpy()
        bounds, obs_bounds = checked_partition(i_host, N, world)      # same verdict on every rank
        p0, p1 = bounds[rank], bounds[rank + 1]
        e0, e1 = obs_bounds[rank], obs_bounds[rank + 1]
        i_loc = (i_t[e0:e1].to(torch.int64) - p0).to(torch.int32)
        shard, failure = None, None
        try:
            shard = engine.Shard(i_loc, j_t[e0:e1], uv_t[e0:e1], p1 - p0, K, W, H, dev, n_visible_total=nV)
            shard.set_params(X_0[p0:p1], C_0, Theta_0, phi_x_0)
        except Exception as e:        # the other ranks are about to enter collectives: tell them first
            failure = e
        if not all_ranks_ok(failure is None, dev):
            if shard is not None:
                shard.close()
            raise failure if failure is not None else RuntimeError('another rank could not set up its shard')

        def gather_X(x_local):
            from .sharding import all_gather_rows
            torch.cuda.synchronize(dev)         # the loop's collectives (another communicator) have drained
            return all_gather_rows(x_local, bounds)
    else:
        shard = engine.Shard(i_t, j_t, uv_t, N, K, W, H, dev)
        shard.set_params(X_0, C_0, Theta_0, phi_x_0)
    return _run_fit(shard, lr, max_iters, (X_mask, C_Theta_mask, phi_x_mask), optimizer, patience, factor, stop_value,
                    stop_diff, print_step, ret_all_iters, gather_X)


def _run_fit(shard, lr, max_iters, masks, optimizer, patience, factor, stop_value, stop_diff, print_step, ret_all_iters,
             gather_X):
    try:
        if optimizer == 'LM':
            fit = LMFit(shard, max_iters, lam=lr, stop_value=stop_value, stop_diff=stop_diff, print_step=print_step)
            fit.ret_all_iters = False
        else:
            fit = SceneFit(shard, lr, max_iters, *masks, optimizer, patience, factor, stop_value, stop_diff, print_step,
                           ret_all_iters)
        fit.run()
        return _pack_results(fit, shard, masks, gather_X)
    finally:
        shard.close()


def fit_scene_sharded(i_local, j_local, uv_local, N_local, K, W, H, lr, max_iters, X_0_local, C_0, Theta_0, phi_x_0,
                      X_mask=True, C_Theta_mask=True, phi_x_mask=True, optimizer='Adam', patience=1000, factor=2.0,
                      stop_value=0.0, stop_diff=1e-12, print_step=1000, ret_all_iters=False, device=None,
                      check_connectivity=True, gather=False, group=None):
    """`fit_scene_sparse` for callers that already hold the scene in pieces: every rank of the (initialised)
    torch.distributed group passes ONLY its own points - local numbers 0 .. N_local - 1, in the order of the
    global numbering (rank 0's points first) - with their observations (sorted by point, device tensors or arrays)
    and initial positions; cameras and phi_x are passed in full by everyone.  Nothing of the size of the whole
    scene is built on any rank.  Same checks and messages as the reference on the WHOLE scene (the counts are
    added up over the ranks, the visibility graphs are merged through the cameras), same optimisation.  Returns
    the reference's list with this rank's block of X (and of its history); gather=True returns all points on
    every rank."""
    dist = torch.distributed
    X_0_local, C_0, Theta_0 = (np.asarray(a, dtype=np.float64) for a in (X_0_local, C_0, Theta_0))
    validate_scene_parameters(X=X_0_local, C=C_0, Theta=Theta_0, phi_x=phi_x_0)
    if X_0_local.shape[0] != N_local or C_0.shape[0] != K:
        raise ValueError(f'Expected X_0_local, C_0 to have shape[0] = {N_local}, {K}, got {X_0_local.shape[0]}, {C_0.shape[0]}')
    dev = torch.device(device) if device is not None else _default_device()
    several = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    as_t = lambda a: a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
    i_t, j_t, uv_t = as_t(i_local), as_t(j_local), as_t(uv_local)
    counts = torch.tensor([int(i_t.numel()), int(N_local)], dtype=torch.int64)
    sizes = [int(N_local)]
    if several:
        host = dist.get_backend(group) == 'gloo'
        c = counts if host else counts.to(dev)
        dist.all_reduce(c, group=group)
        counts = c.cpu()
        sizes = [None] * dist.get_world_size(group)
        dist.all_gather_object(sizes, int(N_local), group=group)
    nV, N = int(counts[0]), int(counts[1])
    check_solvable(nV, N, K, X_mask, C_Theta_mask, phi_x_mask)
    if optimizer not in gd_func_dict and optimizer not in extra_optimizers:
        raise ValueError(f'Unkown name of the optmizer, choose one from {[key for key in gd_func_dict]}')
    if optimizer == 'LM' and not (X_mask and C_Theta_mask and phi_x_mask):
        raise ValueError("optimizer='LM' needs X_mask, C_Theta_mask and phi_x_mask to be true")
    shard, failure = None, None
    try:
        if int(i_t.numel()) == 0 or N_local <= 0:
            raise ValueError('every rank must hold at least one point with an observation')
        shard = engine.Shard(i_t, j_t, uv_t, N_local, K, W, H, dev, n_visible_total=nV, group=group)
        shard.set_params(X_0_local, C_0, Theta_0, phi_x_0)
    except Exception as e:        # the other ranks are about to enter collectives: tell them first
        failure = e
    if not all_ranks_ok(failure is None, dev, group):
        if shard is not None:
            shard.close()
        raise failure if failure is not None else RuntimeError('another rank could not set up its shard')
    if X_mask and C_Theta_mask and check_connectivity:
        n = visibility_graph_components(i_t, j_t, N_local, K, dev, group)      # the same number on every rank
        if n > 1:
            shard.close()
            raise ValueError('Unable to solve the problem, expected the number of connected components of '
                             f'visibility graph to be 1, got {n}')
    gather_X = None
    if gather and several:
        bounds = np.concatenate([[0], np.cumsum(sizes)]).tolist()

        def gather_X(x_local):
            from .sharding import all_gather_rows
            torch.cuda.synchronize(dev)
            return all_gather_rows(x_local, bounds, group)
    return _run_fit(shard, lr, max_iters, (X_mask, C_Theta_mask, phi_x_mask), optimizer, patience, factor, stop_value,
                    stop_diff, print_step, ret_all_iters, gather_X)


def fit_scene(X_pix, W, H, lr, max_iters, X_0, C_0, Theta_0, phi_x_0, X_mask=True, C_Theta_mask=True,
              phi_x_mask=True, optimizer='Adam', patience=1000, factor=2.0, stop_value=0.0, stop_diff=1e-12,
              print_step=1000, ret_all_iters=False, device=None):
    """Bundle adjustment of points X, camera centres C, Euler angles Theta and the shared
    horizontal field of view phi_x on the visible entries of the integer track array
    X_pix (N,K,2) — signature and return value of the reference's fit_scene (p2w:480-500):
    `[X?, C?, Theta?, phi_x?, E[:iters+1]]` followed, with ret_all_iters, by the per-iteration
    history of each free group; a group is present only if its mask is True."""
    N, K = validate_scene_parameters(X=X_0, C=C_0, Theta=Theta_0, phi_x=phi_x_0, X_pix=X_pix, ret_shapes=True)
    if not X_mask and not C_Theta_mask and not phi_x_mask:
        raise ValueError('At least one of X_mask, C_Theta_mask or phi_x_mask have to be true')
    dev = torch.device(device) if device is not None else _default_device()
    oi, oj, uv, _ = engine.visibility_coo(np.asarray(X_pix), dev)
    return fit_scene_sparse(oi, oj, uv, N, K, W, H, lr, max_iters, X_0, C_0, Theta_0, phi_x_0, X_mask,
                            C_Theta_mask, phi_x_mask, optimizer, patience, factor, stop_value, stop_diff,
                            print_step, ret_all_iters, device=dev, _validate=False)


# --------------------------------------------------------------------------------------------- #
# rotation and gauge helpers (p2w:138-140, 254-333)
# --------------------------------------------------------------------------------------------- #
def ryxz(Theta, _validate=True, device=None):
    """Rotation matrices Ry(theta_y) @ Rx(theta_x) @ Rz(theta_z), shape (K, 3, 3) (p2w:138-140)."""
    if _validate:
        validate_scene_parameters(Theta=Theta)
    dev = torch.device(device) if device is not None else _default_device()
    return engine.rotation_matrices(Theta, dev).cpu().numpy()


def get_rotations(R, _validate=True, device=None):
    """Euler angles (K, 3) of rotation matrices, the inverse of ryxz with theta_x in [-pi/2, pi/2]
    (p2w:254-267)."""
    if _validate:
        validate_scene_parameters(R=R)
    dev = torch.device(device) if device is not None else _default_device()
    return engine.rotation_angles(R, dev).cpu().numpy()


def align_scene(X, C, Theta=None, R=None, ret_R=False, _validate=True, device=None):
    """The scene in the frame of camera 0 (its centre at the origin, its rotation the identity):
    returns X_new, C_new, Theta_new and, with ret_R, the new rotation matrices (p2w:270-303)."""
    if Theta is None and R is None:
        raise ValueError('Expected Theta or R to be passed')
    elif Theta is not None and R is not None:
        raise ValueError('Expected only Theta or R to be passed, not both')
    if _validate:
        validate_scene_parameters(X=X, C=C, Theta=Theta, R=R)
    dev = torch.device(device) if device is not None else _default_device()
    out = engine.align_scene(X, C, Theta, R, dev)
    out = [t.cpu().numpy() for t in out]
    return tuple(out) if ret_R else tuple(out[:3])


def normalize_scene(X, C, main_indexes=None, dist_scale=1.0, ret_j=False, _validate=True, device=None):
    """X and C scaled so that the distance between cameras main_indexes = (j0, j1) becomes dist_scale;
    by default camera 0 and the camera farthest from it.  Returns [X_new, C_new] and, with ret_j, j1
    (p2w:306-333)."""
    if _validate:
        validate_scene_parameters(X=X, C=C)
        if main_indexes is not None and len(main_indexes) != 2:
            raise ValueError(f'Expected main_indexes to have length 2, got {len(main_indexes)}')
    dev = torch.device(device) if device is not None else _default_device()
    X_new, C_new, j1 = engine.normalize_scene(X, C, main_indexes, dist_scale, dev)
    returns = [X_new.cpu().numpy(), C_new.cpu().numpy()]
    if ret_j:
        returns.append(j1)
    return returns


def transform(X, C, Theta, phi_x, W, H, delete_back_points=True, delete_boundary_points=False, _validate=True,
              device=None):
    """Model pixels of every point in every camera, shape (N,K,2), NaN behind the camera and
    (optionally) outside the frame (p2w:192-216)."""
    if _validate:
        validate_scene_parameters(X=X, C=C, Theta=Theta, phi_x=phi_x)
    dev = torch.device(device) if device is not None else _default_device()
    out = engine.project(X, C, Theta, phi_x, W, H, None, None, delete_back_points, delete_boundary_points, dev)
    return out.view(X.shape[0], C.shape[0], 2).cpu().numpy()


def error(X_pix, X_model, device=None):
    """Mean squared reprojection error over pairs visible in both arrays, halved (p2w:98-111)."""
    validate_scene_parameters(X_pix=X_pix)
    if X_model.shape != X_pix.shape:
        raise ValueError(f'Expected X_pix and X_model to have the same shape, got {X_pix.shape} != {X_model.shape}')
    dev = torch.device(device) if device is not None else _default_device()
    s, n = engine.reprojection_error(np.asarray(X_pix), np.asarray(X_model), dev)
    return np.float64(s) / (2 * n)


def get_X_from_C_Theta(X_pix, C, Theta, phi_x, W, H, normalize=False, _validate=True, device=None):
    """Linear triangulation of every point from the cameras (p2w:1655-1721): (N,3), NaN for points
    seen fewer than twice or without enough parallax."""
    if _validate:
        N, K = validate_scene_parameters(X_pix=X_pix, C=C, Theta=Theta, phi_x=phi_x, ret_shapes=True)
    else:
        N, K = X_pix.shape[:2]
    dev = torch.device(device) if device is not None else _default_device()
    oi, oj, uv, row_ptr = engine.visibility_coo(np.asarray(X_pix), dev)
    X = engine.triangulate(oi, oj, uv, N, C, Theta, phi_x, W, H, normalize, dev, row_ptr=row_ptr).cpu().numpy()
    if np.isnan(X).all():
        import warnings
        seen_twice = bool((row_ptr[1:] - row_ptr[:-1] >= 2).any())
        warnings.warn('Unable to solve the problem for any point, too little valuable information' if seen_twice
                      else 'Expected each point to be visible at least 2 times', RuntimeWarning)      # p2w:1674, 1709
    return X


def get_C_Theta_from_X(X_pix, X, phi_x, W, H, normalize=False, _validate=True, device=None):
    """DLT resection of every camera from the points (p2w:1724-1808): C (K,3), Theta (K,3), NaN for
    cameras that see fewer than 6 points or whose points do not determine them."""
    if _validate:
        N, K = validate_scene_parameters(X_pix=X_pix, X=X, phi_x=phi_x, ret_shapes=True)
    else:
        N, K = X_pix.shape[:2]
    dev = torch.device(device) if device is not None else _default_device()
    oi, oj, uv, _ = engine.visibility_coo(np.asarray(X_pix), dev)
    C, Theta = engine.resect(oi, oj, uv, K, X, phi_x, W, H, normalize, dev)
    C, Theta = C.cpu().numpy(), Theta.cpu().numpy()
    if np.isnan(C).all():
        import warnings
        warnings.warn('Unable to solve the problem for any camera, too little valuable information', RuntimeWarning)
    return C, Theta


def get_H_diff_matrix(X_pix, normalize=False, W=None, H=None, _validate=True, device=None):
    """Mean squared transfer error of the best homography between every pair of cameras with at
    least 10 common points (p2w:1945-1995): (K, K), symmetric, NaN for the other pairs and on the
    diagonal.  A large value means parallax between the two views."""
    if _validate:
        N, K = validate_scene_parameters(X_pix=X_pix, ret_shapes=True)
        if normalize and (W is None or H is None):
            raise ValueError('With normalize = True W and H have to be passed')
    else:
        N, K = X_pix.shape[:-1]
    dev = torch.device(device) if device is not None else _default_device()
    oi, oj, uv, _ = engine.visibility_coo(np.asarray(X_pix), dev)
    return engine.homography_diff_matrix(oi, oj, uv, N, K, normalize, W, H, 10, dev).cpu().numpy()
