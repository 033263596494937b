"""Synthetic "video orbit" scenes for measurement at sizes the reference's own generator
(p2w:2338-2427; dense, 100 % visible, O(N*K)) cannot reach.

K cameras stand on a circle of radius 10 in the xz-plane and look radially outward; every point
is tracked through `track_len` consecutive frames around a random centre frame, as a keypoint in a
video would be.  Pixels are floor(projection) as int32, the integer track format the reference
consumes; observations outside the frame or behind a camera are dropped.  Output is the COO
list sorted by point then camera — the order numpy.nonzero gives on the dense visibility mask."""
from __future__ import annotations

import numpy as np


def _ryxz(Theta):
    """R = Ry(theta_y) Rx(theta_x) Rz(theta_z) for (K,3) angles (columns x, y, z)."""
    sx, cx = np.sin(Theta[:, 0]), np.cos(Theta[:, 0])
    sy, cy = np.sin(Theta[:, 1]), np.cos(Theta[:, 1])
    sz, cz = np.sin(Theta[:, 2]), np.cos(Theta[:, 2])
    R = np.empty((Theta.shape[0], 3, 3))
    R[:, 0, 0] = cy * cz + sy * sx * sz
    R[:, 0, 1] = -cy * sz + sy * sx * cz
    R[:, 0, 2] = sy * cx
    R[:, 1, 0] = cx * sz
    R[:, 1, 1] = cx * cz
    R[:, 1, 2] = -sx
    R[:, 2, 0] = -sy * cz + cy * sx * sz
    R[:, 2, 1] = sy * sz + cy * sx * cz
    R[:, 2, 2] = cy * cx
    return R


def orbit_cameras(K, seed=0):
    """Camera centres, angles and rotations of the orbit (identical for every shard of a scene)."""
    rng = np.random.default_rng([seed, 0xCA])
    alpha = 2.0 * np.pi * np.arange(K) / K
    alpha = np.where(alpha > np.pi, alpha - 2.0 * np.pi, alpha)          # wrap into (-pi, pi]
    C = 10.0 * np.stack([np.sin(alpha), np.zeros(K), np.cos(alpha)], axis=1)
    Theta = np.stack([np.zeros(K), alpha, np.zeros(K)], axis=1)
    Theta = Theta + rng.normal(0.0, 0.01, (K, 3))
    C = C + rng.normal(0.0, 0.02, (K, 3))
    return C, Theta, _ryxz(Theta)


def orbit_scene(N, K, track_len=10, seed=0, W=1920, H=1080, phi_x=1.2, part=0, parts=1, stride=1, centres=None):
    """Points `part` of `parts` (contiguous ranges of the N points) of the orbit scene.

    Returns a dict with: i (local point index), j, xy (int32 pixels), uv (= xy + 0.5),
    N_local, point_offset, K, W, H, truth (X, C, Theta, phi_x) and a perturbed initial guess
    (X_0, C_0, Theta_0, phi_x_0).  `centres` (optional, one frame number per point of this part) replaces
    the random centre frames of the tracks."""
    C, Theta, R = orbit_cameras(K, seed)
    p0, p1 = (N * part) // parts, (N * (part + 1)) // parts
    n = p1 - p0
    rng = np.random.default_rng([seed, 0x9017, part, parts])
    k_c = rng.integers(0, K, n)
    if centres is not None:
        k_c = np.asarray(centres, dtype=np.int64) % K
    z = rng.uniform(4.0, 8.0, n)
    t = np.tan(phi_x / 2.0)
    a = rng.uniform(-0.5, 0.5, n) * t
    b = rng.uniform(-0.5, 0.5, n) * t * H / W
    local = np.stack([a * z, b * z, z], axis=1)
    X = C[k_c] + np.einsum('nab,nb->na', R[k_c], local)

    offs = (np.arange(track_len) - track_len // 2) * stride
    jj = (k_c[:, None] + offs[None, :]) % K                               # (n, L)
    jj.sort(axis=1)
    ii = np.repeat(np.arange(n), track_len).reshape(n, track_len)
    d = X[:, None, :] - C[jj]                                             # (n, L, 3)
    q = np.einsum('nlab,nla->nlb', R[jj], d)
    f = W / 2.0 / t
    with np.errstate(divide='ignore', invalid='ignore'):
        px = f * q[..., 0] / q[..., 2] + W / 2.0
        py = f * q[..., 1] / q[..., 2] + H / 2.0
    ok = (q[..., 2] > 0) & (px >= 0) & (px < W) & (py >= 0) & (py < H)
    # drop duplicate cameras of one point (only possible when track_len*stride > K)
    dup = np.zeros_like(ok)
    dup[:, 1:] = jj[:, 1:] == jj[:, :-1]
    ok &= ~dup
    xy = np.stack([np.floor(px[ok]), np.floor(py[ok])], axis=1).astype(np.int32)
    obs_i = ii[ok].astype(np.int32)
    obs_j = jj[ok].astype(np.int32)

    rng0 = np.random.default_rng([seed, 0x1217, part, parts])
    rngc = np.random.default_rng([seed, 0x1218])
    return dict(i=obs_i, j=obs_j, xy=xy, uv=xy.astype(np.float64) + 0.5, N_local=n, point_offset=p0,
                N=N, K=K, W=W, H=H, X=X, C=C, Theta=Theta, phi_x=phi_x,
                X_0=X + rng0.normal(0.0, 0.02, X.shape),
                C_0=C + rngc.normal(0.0, 0.02, C.shape),
                Theta_0=Theta + rngc.normal(0.0, 0.005, Theta.shape),
                phi_x_0=phi_x * 1.01)
