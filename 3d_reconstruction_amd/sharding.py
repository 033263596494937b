"""Partitioning of a scene over GPUs and the graph check that precedes a fit.

Observations are independent and the energy is a plain sum, so the path shards by POINT
ranges: a rank owns a contiguous block of points, all their observations, their gradients and
optimiser moments.  Cameras and phi_x are replicated; the camera-side sums and the energy are
the one thing exchanged (one sum over the ranks per evaluation, engine.Shard.eval): by RCCL on the
caller's stream (StreamAllReduce; torch.distributed's own call if that cannot be set up), or, when
BA_NODE_EXCHANGE=1 asks for it, inside a node by the library's own exchange kernel (NodeExchange)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import check, load


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


def checked_partition(obs_i, N, world):
    """partition_points for a fit over `world` ranks, with the checks that have to give the same verdict on
    every rank before any of them builds its shard (a rank that raised alone would leave the others waiting
    in their first collective): observations sorted by point, and no rank left without points or
    observations.  Every rank evaluates all ranks' ranges from the same input, so all raise or none does."""
    obs_i = np.asarray(obs_i)
    if obs_i.size > 1 and bool((obs_i[1:] < obs_i[:-1]).any()):
        raise ValueError('observations must be sorted by point index')
    point_bounds, obs_bounds = partition_points(obs_i, N, world)
    for r in range(world):
        if point_bounds[r + 1] <= point_bounds[r] or obs_bounds[r + 1] <= obs_bounds[r]:
            raise ValueError(f'Unable to split {N} points with {obs_i.size} observations over {world} ranks: rank {r} '
                             'would hold no observations; use fewer ranks')
    return point_bounds, obs_bounds


def all_ranks_ok(flag, device=None, group=None):
    """True if `flag` is true on every rank of the group (one MIN all-reduce; no group: the flag itself)."""
    dist = torch.distributed
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return bool(flag)
    t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64)
    if dist.get_backend(group) != 'gloo' and device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.item() > 0.5)


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


class StreamAllReduce:
    """Sum of an fp64 device tensor over the ranks of a group, queued on the CALLER's stream.

    `torch.distributed.all_reduce` runs the RCCL kernel on a stream of its own, i.e. two event
    hand-overs between streams per call (~10 us on the critical path of an iteration whose kernels
    take ~75 us).  Here `ncclAllReduce` of the RCCL library that torch has already loaded is called
    directly, on the stream the neighbouring kernels use.  The communicator is created from a unique
    id made on the group's first rank and broadcast through torch.distributed; a trial sum is
    compared with torch.distributed's before the object is used.  `StreamAllReduce.create` returns
    None (callers then stay with torch.distributed) whenever any of that is not possible."""

    last_created = None      # the latest object `create` handed out in this process (tests look at it)
    NCCL_FLOAT64, NCCL_SUM = 8, 0
    NCCL_DTYPE = {torch.float64: 8, torch.int32: 2, torch.int64: 4}      # ncclDataType_t
    NCCL_OP = {'sum': 0, 'max': 2, 'min': 3}                             # ncclRedOp_t

    def __init__(self, lib, comm):
        self._lib, self._comm = lib, comm

    def __call__(self, t, stream, op='sum'):
        """In-place reduction (sum / max / min) of a float64, int32 or int64 device tensor over the ranks."""
        import ctypes as C
        p = C.c_void_p(t.data_ptr())
        st = stream if isinstance(stream, C.c_void_p) else C.c_void_p(stream)
        rc = self._lib.ncclAllReduce(p, p, C.c_size_t(t.numel()), self.NCCL_DTYPE[t.dtype], self.NCCL_OP[op], self._comm, st)
        if rc != 0:
            raise RuntimeError(f'ncclAllReduce failed with status {rc}')

    def broadcast(self, t, root, stream):
        """Rank `root`'s values of a device tensor to every rank, in place, on the caller's stream."""
        import ctypes as C
        p = C.c_void_p(t.data_ptr())
        st = stream if isinstance(stream, C.c_void_p) else C.c_void_p(stream)
        rc = self._lib.ncclBroadcast(p, p, C.c_size_t(t.numel()), self.NCCL_DTYPE[t.dtype], int(root), self._comm, st)
        if rc != 0:
            raise RuntimeError(f'ncclBroadcast failed with status {rc}')

    @classmethod
    def create(cls, group, device):
        import ctypes as C
        import os
        dist = torch.distributed
        try:
            if dist.get_backend(group) != 'nccl' or os.environ.get('BA_TORCH_ALLREDUCE'):
                return None
            path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
            ok = torch.tensor([1.0 if os.path.exists(path) else 0.0], dtype=torch.float64, device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if float(ok.item()) < 0.5:
                return None
            lib = C.CDLL(path)

            class UniqueId(C.Structure):
                _fields_ = [('internal', C.c_byte * 128)]

            lib.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
            lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
            lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
            lib.ncclBroadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
            for fn in (lib.ncclGetUniqueId, lib.ncclCommInitRank, lib.ncclAllReduce, lib.ncclBroadcast):
                fn.restype = C.c_int
            rank, world = dist.get_rank(group), dist.get_world_size(group)
            uid = UniqueId()
            if rank == 0 and lib.ncclGetUniqueId(C.byref(uid)) != 0:
                uid = UniqueId()        # all zeros: the trial below fails on every rank
            raw = torch.tensor(list(bytes(uid)), dtype=torch.int32, device=device)
            dist.broadcast(raw, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            uid = UniqueId.from_buffer_copy(bytes(int(v) & 0xff for v in raw.tolist()))
            comm = C.c_void_p()
            with torch.cuda.device(device):
                rc = lib.ncclCommInitRank(C.byref(comm), world, uid, rank)
            good = torch.tensor([1.0 if rc == 0 and comm.value else 0.0], dtype=torch.float64, device=device)
            dist.all_reduce(good, op=dist.ReduceOp.MIN, group=group)
            if float(good.item()) < 0.5:
                return None
            obj = cls(lib, comm)
            # trial: the same sum both ways
            a = torch.arange(1, 1025, dtype=torch.float64, device=device) * (rank + 1)
            b = a.clone()
            with torch.cuda.device(device):
                obj(a, torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()        # never two collectives of two communicators in flight on one GPU
            dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group)
            ok = torch.equal(a, b)
            torch.cuda.synchronize(device)
            # the other two calls the fit loop makes: minimum of int32, broadcast from the first rank
            m = torch.tensor([rank + 5, 9 - rank], dtype=torch.int32, device=device)
            c = torch.full((7,), float(rank + 1), dtype=torch.float64, device=device)
            with torch.cuda.device(device):
                obj(m, torch.cuda.current_stream().cuda_stream, op='min')
                obj.broadcast(c, 0, torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
            ok = ok and m.tolist() == [5, 10 - world] and c.tolist() == [1.0] * 7
            same = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=device)
            dist.all_reduce(same, op=dist.ReduceOp.MIN, group=group)
            torch.cuda.synchronize(device)
            if float(same.item()) > 0.5:
                cls.last_created = obj
                return obj
            return None
        except Exception:
            return None


class NodeExchange:
    """Sum of a short fp64 device tensor over the ranks of ONE node by the library's own kernel
    (`ba_exchange_sum`: every GPU publishes its vector in a window the others have mapped and adds the
    windows up in rank order), queued on the caller's stream.  For the camera-block sum of an iteration
    (~100 KB) a collective library is bound by its latency; this is one kernel with two trips over
    xGMI.  `create` exchanges the window handles through torch.distributed, runs a trial sum against
    torch.distributed's and returns None - callers stay with the collective library - whenever a
    step fails on any rank (ranks on several nodes, no peer mapping, a wrong trial result)."""

    created = 0          # exchanges set up successfully in this process

    _shared = {}         # (device index, capacity) -> (process group, exchange or None)

    def __init__(self, lib, handle, capacity):
        self._lib, self._handle, self.capacity = lib, handle, capacity
        self._last_stream = None

    def __call__(self, t, stream):
        st = stream if isinstance(stream, C.c_void_p) else C.c_void_p(stream)
        if self._last_stream is not None and self._last_stream != st.value:
            # a call starts only after the previous one has finished on this GPU: guaranteed by the stream's
            # order as long as the stream stays the same
            torch.cuda.synchronize(t.device)
        self._last_stream = st.value
        check(self._lib.ba_exchange_sum(self._handle, C.c_void_p(t.data_ptr()), t.numel(), st))

    @classmethod
    def shared(cls, group, device, capacity):
        """The exchange of this process for (group, device), set up on first use (None if that failed: not
        tried again).  It lives until the process ends: a rank cannot know when the others have finished
        reading its window, so the window is never freed while the group is alive.  Every rank of the group
        must ask for it at the same point of the program."""
        import os
        # opt-in: the default path for the sums over the ranks is RCCL on the compute stream (StreamAllReduce)
        if os.environ.get('BA_NODE_EXCHANGE') != '1':
            return None
        dist = torch.distributed
        pg = group if group is not None else dist.distributed_c10d._get_default_group()
        key = (torch.device(device).index, int(capacity))
        hit = cls._shared.get(key)
        if hit is None or hit[0] is not pg:
            hit = (pg, cls.create(group, device, capacity))
            cls._shared[key] = hit
        return hit[1]

    def check(self):
        """Raises if a call gave up waiting for another rank (read from host memory: no sync)."""
        check(self._lib.ba_exchange_status(self._handle))

    def close(self):
        """Frees the window.  Only safe once every rank has finished its last sum (a barrier after a device
        synchronisation on every rank): the others read this rank's window from inside their kernels."""
        if self._handle is not None:
            self._lib.ba_exchange_destroy(self._handle)
            self._handle = None

    @staticmethod
    def _all_ok(flag, group, device, host):
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device='cpu' if host else device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN, group=group)
        return bool(t.item() > 0.5)

    @classmethod
    def create(cls, group, device, capacity):
        import os
        import socket
        dist = torch.distributed
        host = dist.get_backend(group) == 'gloo'
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        handle, obj = C.c_void_p(), None
        try:
            lib = load()
            names = [None] * world
            dist.all_gather_object(names, socket.gethostname(), group=group)
            ok = world <= 16 and len(set(names)) == 1
            mine = (C.c_byte * 64)()
            if ok:
                with torch.cuda.device(device):
                    ok = lib.ba_exchange_create(rank, world, int(capacity), C.byref(handle)) == 0 and \
                        lib.ba_exchange_export(handle, mine) == 0
            handles = [None] * world
            dist.all_gather_object(handles, bytes(mine) if ok else None, group=group)
            ok = ok and all(h is not None for h in handles)
            if ok:
                with torch.cuda.device(device):
                    for r in range(world):
                        if r != rank and lib.ba_exchange_import(handle, r, (C.c_byte * 64).from_buffer_copy(handles[r])) != 0:
                            ok = False
                            break
            if cls._all_ok(ok, group, device, host):         # also the barrier before the first sum
                obj = cls(lib, handle, int(capacity))
                lib.ba_exchange_set_wait(handle, 0.5)
                n = min(int(capacity), 3000)
                a = torch.arange(1, n + 1, dtype=torch.float64, device=device) * (rank + 1) / 7.0
                want = a.cpu() if host else a.clone()
                dist.all_reduce(want, op=dist.ReduceOp.SUM, group=group)
                # 40 calls queued back to back, as the fit loop queues them (no host sync in between: the
                # two slots are reused under GPU-side ordering only), each checked against the collective's sum
                outs, b = [], a
                try:                                 # a failure of this rank must not keep it from the collectives below
                    with torch.cuda.device(device):
                        first = a.clone()            # one call on its own: a rank that cannot see the others' flags
                        obj(first, torch.cuda.current_stream().cuda_stream)       # is found out after one wait
                        torch.cuda.synchronize(device)
                        if lib.ba_exchange_status(handle) != 0:
                            raise RuntimeError('no answer from the other ranks')
                        for k in range(40):
                            b = a * float(k + 1)
                            obj(b, torch.cuda.current_stream().cuda_stream)
                            outs.append(b)
                    torch.cuda.synchronize(device)
                    good = lib.ba_exchange_status(handle) == 0 and len(outs) == 40
                    for k, o in enumerate(outs):     # rank order against the collective's order: equal up to rounding
                        good = good and bool(torch.allclose(o.cpu(), want.cpu() * float(k + 1), rtol=1e-13, atol=0.0))
                except Exception:
                    good = False
                good = cls._all_ok(good, group, device, host)
                if good:
                    same = b.clone() if not host else b.cpu()
                    dist.broadcast(same, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                    good = bool(torch.equal(same.cpu(), b.cpu()))       # identical bits on every rank
                    good = cls._all_ok(good, group, device, host)
                if good:
                    lib.ba_exchange_set_wait(handle, 20.0)
                    cls.created += 1
                    return obj
        except Exception:
            pass
        if obj is not None:
            obj.close()
        elif handle:
            try:
                load().ba_exchange_destroy(handle)
            except Exception:
                pass
        return None
