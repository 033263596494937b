"""Partitioning of a scene over GPUs and the graph check that precedes a fit.

Observations are independent and the energy is a plain sum, so the path shards by POINT
ranges: a rank owns a contiguous block of points, all their observations, their gradients and
optimiser moments.  Cameras and phi_x are replicated; the camera-side sums and the energy are
the one thing exchanged (one all-reduce per evaluation, engine.Shard.eval)."""
from __future__ import annotations

import numpy as np
import torch


def partition_points(obs_i, N, world):
    """Split points 0..N-1 into `world` contiguous ranges with (nearly) equal observation
    counts.  `obs_i` is the point index of every observation, sorted ascending.
    Returns (point_bounds, obs_bounds), each of length world + 1."""
    obs_i = np.asarray(obs_i)
    nV = obs_i.size
    point_bounds, obs_bounds = [0], [0]
    for r in range(1, world):
        target = (nV * r) // world
        if target <= 0:
            e = 0
        elif target >= nV:
            e = nV
        elif obs_i[target - 1] != obs_i[target]:
            e = target
        else:                      # inside a point's run: move to the start of the next point
            e = int(np.searchsorted(obs_i, obs_i[target], side='right'))
        e = max(e, obs_bounds[-1])
        point_bounds.append(int(obs_i[e]) if e < nV else N)
        obs_bounds.append(e)
    point_bounds.append(N)
    obs_bounds.append(nV)
    return point_bounds, obs_bounds


def connected_components_count(i, j, N, K):
    """Number of connected components of the bipartite point/camera visibility graph with
    N + K nodes (the reference builds the dense adjacency, p2w:519-523; the sparse form gives
    the same count in O(|V|))."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    i = np.asarray(i, dtype=np.int64)
    j = np.asarray(j, dtype=np.int64) + N
    g = coo_matrix((np.ones(i.size, dtype=np.int8), (i, j)), shape=(N + K, N + K))
    return int(connected_components(g, directed=False, return_labels=False))


def allreduce_sum_(t, group=None):
    """In-place sum over ranks (RCCL on GPUs; a no-op without an initialised process group)."""
    if torch.distributed.is_available() and torch.distributed.is_initialized() \
            and torch.distributed.get_world_size(group) > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, group=group)
    return t


def all_gather_rows(x_local, point_bounds, group=None):
    """Concatenate the per-rank row blocks of a (n_local, 3) tensor into the full (N, 3) tensor
    on every rank."""
    if x_local.is_cuda and torch.distributed.get_backend(group) == 'gloo':   # stage through the host
        return all_gather_rows(x_local.cpu(), point_bounds, group).to(x_local.device)
    world = len(point_bounds) - 1
    sizes = [point_bounds[r + 1] - point_bounds[r] for r in range(world)]
    width = x_local.shape[1]
    m = max(sizes)
    padded = x_local.new_zeros((m, width))
    padded[:x_local.shape[0]] = x_local
    parts = [x_local.new_zeros((m, width)) for _ in range(world)]
    torch.distributed.all_gather(parts, padded, group=group)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0)
