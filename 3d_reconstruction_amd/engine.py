"""Device-side state of one bundle-adjustment shard and the kernels that act on it.

PyTorch is used for device memory, streams and (multi-GPU) `torch.distributed`; every
computation goes through the C ABI of libba_mi355x.so."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .sharding import NodeExchange, StreamAllReduce, allreduce_sum_

F64 = torch.float64


def _take_rows(t, idx):
    """t[idx] for a 2-D tensor, column by column.  torch's row gather (vectorized_gather_kernel) of this
    ROCm build returns wrong rows for large inputs - measured: an (80 M, 2) float64 tensor, wrong from
    row 12 891 136 on, while the 1-D indexing used here is correct at that size."""
    if t.dim() == 1:
        return t[idx]
    return torch.stack([t[:, c][idx] for c in range(t.shape[1])], dim=1)


def visibility_components(oi, oj, N, K, device, want_labels=False):
    """Connected components of the bipartite visibility graph of a COO list (ba_visibility_components):
    (components with a camera, components of points alone[, camera labels (K,) int32 device tensor])."""
    lib = _lib.load()
    dev = torch.device(device)
    oi, oj = _dev_tensor(oi, torch.int32, dev), _dev_tensor(oj, torch.int32, dev)
    labels = torch.empty(int(K), dtype=torch.int32, device=dev) if want_labels else None
    n_cam, n_pts = C.c_int64(0), C.c_int64(0)
    with torch.cuda.device(dev):
        check(lib.ba_visibility_components(int(N), int(K), int(oi.numel()), ptr(oi), ptr(oj), ptr(labels), C.byref(n_cam),
                                           C.byref(n_pts), stream_ptr()))
    return (int(n_cam.value), int(n_pts.value), labels) if want_labels else (int(n_cam.value), int(n_pts.value))


def _dev_tensor(a, dtype, device):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dtype)


def visibility_coo(X_pix, device):
    """Visibility set of a dense (N,K,2) integer track array, extracted on the GPU in the
    order `numpy.nonzero(X_pix[:,:,0] >= 0)` produces (p2w:506-508, 531).
    Returns device tensors (i int32[V], j int32[V], uv float64[V,2])."""
    lib = _lib.load()
    pix = _dev_tensor(X_pix, torch.int32, device)
    N, K = int(pix.shape[0]), int(pix.shape[1])
    row_ptr = torch.empty(N + 1, dtype=torch.int64, device=device)
    nV = C.c_int64(0)
    with torch.cuda.device(device):
        check(lib.ba_visibility_count(ptr(pix), N, K, ptr(row_ptr), C.byref(nV), stream_ptr()))
        n = int(nV.value)
        oi = torch.empty(n, dtype=torch.int32, device=device)
        oj = torch.empty(n, dtype=torch.int32, device=device)
        uv = torch.empty((n, 2), dtype=F64, device=device)
        check(lib.ba_visibility_fill(ptr(pix), N, K, ptr(row_ptr), ptr(oi), ptr(oj), ptr(uv), stream_ptr()))
    return oi, oj, uv, row_ptr


class Shard:
    """One GPU's share of a scene: a contiguous range of points with all their observations,
    and a replica of every camera.

    Parameter storage: X (N,3); cam = [C (3K) | Theta (3K) | phi_x (1)] in one buffer so that the
    camera-side update is one launch.  Reduction buffer: red = [gC (3K) | gTheta (3K) | gphi (1) |
    sum r^2 (1)], the only thing that crosses GPUs (one all-reduce per evaluation)."""

    def __init__(self, oi, oj, uv, N, K, W, H, device, n_visible_total=None, group=None, reorder=True):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type == 'cuda' and self.device.index is None:     # 'cuda' means the current device, not GPU 0
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.N, self.K, self.W, self.H = int(N), int(K), float(W), float(H)
        oi = _dev_tensor(oi, torch.int32, self.device)
        oj = _dev_tensor(oj, torch.int32, self.device)
        uv = _dev_tensor(uv, F64, self.device).view(-1, 2)
        self.perm = None        # engine row -> caller's point index (None: same order)
        if oi.numel() > 0:
            lo_i, hi_i, lo_j, hi_j = (int(v) for v in (oi.min(), oi.max(), oj.min(), oj.max()))
            if lo_i < 0 or hi_i >= self.N or lo_j < 0 or hi_j >= self.K:
                raise _lib.BAError('observation index out of range')
        if reorder and oi.numel() > 0:
            oi, oj, uv = self._order_points_by_first_camera(oi, oj, uv)
        self.nV_local = int(oi.numel())
        self.nV = int(n_visible_total) if n_visible_total is not None else self.nV_local
        self.group = group
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self.lib.ba_problem_create(self.N, self.K, self.nV_local, ptr(oi), ptr(oj), ptr(uv), self.W,
                                             self.H, self.device.index, C.byref(handle)))
        self.handle = handle
        K3 = 3 * self.K
        z = lambda *s: torch.zeros(*s, dtype=F64, device=self.device)
        # Current parameters.  X and cam are REBOUND, not overwritten, by the optimiser step
        # (step_update returns new tensors, adopt() binds them), so a tensor a caller kept - the
        # snapshot the LR schedule rolls back to - stays valid without a copy.
        # The points may carry a pending gauge record: the true points are then
        # ((X_buf - C0) @ R0) * factor; the fused pass applies it while loading, `X` applies it
        # on demand.
        self._X_buf = z(self.N, 3)
        self._X_gauge = None
        self._cam_buf = z(2 * K3 + 1)
        self._next_rec = None       # gauge record of the next merged camera step (its word 14 already zero)
        self.gX = z(self.N, 3)
        self.red = z(2 * K3 + 2)
        self.gauge = z(16)
        self.R_gauge = z(self.K, 9)
        self._table_cam = None      # the cam tensor the handle's camera table was built from
        # hot-path plumbing: one device, one stream, pointers of the fixed buffers converted once
        torch.cuda.set_device(self.device)
        self.stream = stream_ptr()
        self.distributed = torch.distributed.is_available() and torch.distributed.is_initialized() and \
            torch.distributed.get_world_size(group) > 1
        # RCCL sums device tensors in place; with a gloo group (rehearsals of the multi-rank path on
        # one GPU) the buffers are staged through the host explicitly
        self._stage_reduce = self.distributed and torch.distributed.get_backend(group) == 'gloo'
        self._stream_allreduce = False     # not yet tried; then a StreamAllReduce or None (torch.distributed)
        self._node_exchange = False        # not yet tried; then a NodeExchange (short sums inside one node) or None
        self._bind_gradient(self.gX, self.red)
        self._E_host = C.c_double(0.0)
        # Several GPUs: the sums a deferred pass leaves for the camera step (camera-frame sums, per-tile partial
        # sums) live in one tensor of this class, summed over the ranks with one collective.
        # (created by the first deferred pass: a collective, every rank gets there in the same iteration)
        self.raw = None

    def _bind_gradient(self, gX, red):
        """Makes (gX, red = [gC | gTheta | gphi | sum r^2]) the buffers the passes write."""
        K3 = 3 * self.K
        self.gX, self.red = gX, red
        self.gC = red[:K3].view(self.K, 3)
        self.gTheta = red[K3:2 * K3].view(self.K, 3)
        self.gphi = red[2 * K3:2 * K3 + 1]
        self.E_sum = red[2 * K3 + 1:]
        self._out_ptrs = [t.data_ptr() for t in (self.E_sum, self.gX, self.gC, self.gTheta, self.gphi)]

    @property
    def X(self):
        """Current points (N,3) in engine order; a pending gauge record is applied first."""
        if self._X_gauge is not None:
            out = torch.empty_like(self._X_buf)
            check(self.lib.ba_gauge_points(self.N, self._X_buf.data_ptr(), out.data_ptr(), self._X_gauge.data_ptr(),
                                           self.stream))
            self._X_buf, self._X_gauge = out, None
        return self._X_buf

    @X.setter
    def X(self, value):
        self._X_buf, self._X_gauge = value, None

    @property
    def cam(self):
        """Current [C | Theta | phi_x]; a scale of the gauge fix that is still pending on the centres (the merged
        camera step leaves it to its consumers and hangs the record on the tensor, `_ba_gauge`) is applied first."""
        rec = getattr(self._cam_buf, '_ba_gauge', None)
        if rec is None:
            return self._cam_buf
        # the scaled copy is kept beside the raw tensor; the raw tensor stays the current one (`_cam_buf` is what the
        # table bookkeeping and the fused pass compare by identity), so reading this property changes no state that
        # the kernels' routes depend on
        scaled = getattr(self._cam_buf, '_ba_scaled', None)
        if scaled is None:
            scaled = torch.empty_like(self._cam_buf)
            check(self.lib.ba_scale_centres(self.K, self._cam_buf.data_ptr(), rec.data_ptr(), scaled.data_ptr(), self.stream))
            self._cam_buf._ba_scaled = scaled
        return scaled

    @cam.setter
    def cam(self, value):
        self._cam_buf = value

    @property
    def cam_raw(self):
        """The current camera tensor as it is (its centres may lack a pending scale, see `cam`)."""
        return self._cam_buf

    def snapshot(self):
        """The current parameters as references (nothing is copied; current tensors are never
        written in place by the optimiser steps)."""
        return (self._X_buf, self._X_gauge, self._cam_buf)

    def restore(self, snap):
        self._X_buf, self._X_gauge, self._cam_buf = snap

    @property
    def can_fuse(self):
        """The observation pass can apply the point update itself (eval_update)."""
        return bool(self.lib.ba_can_fuse_update(self.handle))

    def _shared_sum_buffer(self):
        if self.raw is None:
            cap = C.c_int64(0)
            check(self.lib.ba_sum_buffer_size(self.handle, C.byref(cap)))
            t = torch.tensor([float(cap.value)], dtype=F64)
            if torch.distributed.get_backend(self.group) != 'gloo':
                t = t.to(self.device)
            self._use_communicator('torch')
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=self.group)
            cap = int(t.item())
            # [camera-frame sums (6K) | sum r^2, sum m.r]: what one sum over the ranks carries per iteration
            self.raw = torch.zeros(6 * self.K + 2 * cap, dtype=F64, device=self.device)
            check(self.lib.ba_set_sum_buffer(self.handle, self.raw.data_ptr(), cap))
        return self.raw

    def all_ranks_agree(self, flag):
        """True if `flag` is true on every rank of the group."""
        t = torch.tensor([1.0 if flag else 0.0], dtype=F64)
        if torch.distributed.get_backend(self.group) != 'gloo':
            t = t.to(self.device)
        self._use_communicator('torch')
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN, group=self.group)
        return bool(t.item() > 0.5)

    # longest sum that goes through the node exchange (doubles): the camera-block sums of an iteration and of
    # a J^T J pass at a few thousand cameras; longer ones (the Schur complement) are bandwidth-bound and stay
    # with the collective library
    NODE_EXCHANGE_CAPACITY = 131072

    # Two communicators serve a fit on RCCL: the library's own (StreamAllReduce: every collective of the loops, on
    # the compute stream) and torch.distributed's (set-up and result gathering, on a stream of torch's).  Collectives
    # of two communicators must never be in flight together on one GPU (their kernels can start in different orders
    # on different ranks and wait for each other for ever), so whoever changes from one to the other first waits for
    # the device to drain.
    def _use_communicator(self, which):
        if getattr(self, '_communicator_in_use', which) != which and not self._stage_reduce:
            torch.cuda.synchronize(self.device)
        self._communicator_in_use = which

    def _own_collectives(self):
        """The library's communicator on the compute stream (None: torch.distributed carries everything)."""
        if self._stream_allreduce is False:
            self._use_communicator('torch')         # its set-up talks through torch.distributed
            self._stream_allreduce = None if self._stage_reduce else StreamAllReduce.create(self.group, self.device)
        return self._stream_allreduce

    def _sum_over_ranks(self, t):
        if self._node_exchange is False:         # first use: every rank gets here in the same iteration
            self._use_communicator('torch')
            self._node_exchange = NodeExchange.shared(self.group, self.device, self.NODE_EXCHANGE_CAPACITY)
        if self._node_exchange is not None and t.numel() <= self._node_exchange.capacity:
            self._node_exchange(t, self.stream)
            return
        own = self._own_collectives()
        if own is not None:
            self._use_communicator('own')
            own(t, self.stream)
            return
        self._use_communicator('torch')
        if self._stage_reduce:
            host = t.cpu()
            torch.distributed.all_reduce(host, op=torch.distributed.ReduceOp.SUM, group=self.group)
            t.copy_(host)
        else:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, group=self.group)

    def _min_over_ranks(self, t):
        """Elementwise minimum of a device tensor over the ranks of the group, in place."""
        dist = torch.distributed
        own = self._own_collectives()
        if own is not None:
            self._use_communicator('own')
            own(t, self.stream, op='min')
            return
        self._use_communicator('torch')
        if self._stage_reduce:
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MIN, group=self.group)
            t.copy_(host)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)

    def _broadcast_first(self, t):
        """The first rank's values of a device tensor to every rank of the group, in place."""
        dist = torch.distributed
        own = self._own_collectives()
        if own is not None:
            self._use_communicator('own')
            own.broadcast(t, 0, self.stream)
            return
        self._use_communicator('torch')
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        if self._stage_reduce:
            host = t.cpu()
            dist.broadcast(host, src=src, group=self.group)
            t.copy_(host)
        else:
            dist.broadcast(t, src=src, group=self.group)

    @property
    def rank_sum_path(self):
        """How the short sums over the ranks travel (known after the first one)."""
        if not self.distributed:
            return 'single rank'
        if self._node_exchange:
            return 'node exchange kernel (xGMI windows)'
        if self._node_exchange is False:
            return 'not used yet'
        if self._stream_allreduce:
            return 'RCCL all-reduce on the compute stream'
        return 'gloo through the host' if self._stage_reduce else 'torch.distributed all-reduce'

    # views of the current camera buffer
    @property
    def C(self):
        return self.cam[:3 * self.K].view(self.K, 3)

    @property
    def Theta(self):
        return self.cam[3 * self.K:6 * self.K].view(self.K, 3)

    @property
    def phi(self):
        return self.cam[6 * self.K:]

    @property
    def table_valid(self):
        return self._table_cam is self._cam_buf

    def close(self):
        self._node_exchange = None          # the exchange belongs to the process, not to the shard
        if getattr(self, 'handle', None):
            self.lib.ba_problem_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _order_points_by_first_camera(self, oi, oj, uv):
        """Internal point order: ascending first observing camera, so that consecutive points are
        seen by the same few cameras and the observation list cuts into tiles with short camera
        lists (a keypoint track of a video lives in consecutive frames).  The parameters X and
        everything indexed by point are kept in this order on the device; `points_to_caller`
        restores the caller's order.  Nothing is permuted when the input already has the property.
        (ba_order_points_by_first_camera: key build, stable radix sort and the rewrite of the list on the device.)"""
        nV = int(oi.numel())
        perm = torch.empty(self.N, dtype=torch.int32, device=self.device)
        new_i, new_j, new_uv = torch.empty_like(oi), torch.empty_like(oj), torch.empty_like(uv)
        identity = C.c_int(0)
        with torch.cuda.device(self.device):
            check(self.lib.ba_order_points_by_first_camera(self.N, self.K, nV, ptr(oi), ptr(oj), ptr(uv), ptr(perm),
                                                           ptr(new_i), ptr(new_j), ptr(new_uv), C.byref(identity),
                                                           stream_ptr()))
        if identity.value:
            return oi, oj, uv
        self.perm = perm.to(torch.int64)
        return new_i, new_j, new_uv

    def points_to_engine(self, a):
        """(N, ...) array in the caller's point order -> device tensor in engine order."""
        t = _dev_tensor(a, F64, self.device)
        return t if self.perm is None else _take_rows(t, self.perm)

    def points_to_caller(self, t):
        """Device tensor indexed by engine point order -> same data in the caller's order."""
        if self.perm is None:
            return t
        out = torch.empty_like(t)
        if t.dim() == 1:
            out[self.perm] = t
        else:                       # column by column, see _take_rows
            for c in range(t.shape[1]):
                out[:, c][self.perm] = t[:, c]
        return out

    # ---- parameters -------------------------------------------------------------------------
    def set_params(self, X, C_, Theta, phi_x):
        self.X = self.points_to_engine(X).clone()
        cam = torch.empty_like(self._cam_buf)
        K3 = 3 * self.K
        cam[:K3] = _dev_tensor(C_, F64, self.device).reshape(-1)
        cam[K3:2 * K3] = _dev_tensor(Theta, F64, self.device).reshape(-1)
        cam[2 * K3:] = float(phi_x)
        self.cam = cam
        self._table_cam = None

    def adopt(self, X, cam, X_gauge=None):
        """Make X / cam (tensors returned by step_update / eval_update + camera_update) current."""
        self._X_buf, self._X_gauge, self._cam_buf = X, X_gauge, cam

    def invalidate_table(self):
        """Call after writing into the current camera buffer in place."""
        self._table_cam = None

    # ---- kernels ----------------------------------------------------------------------------
    def eval(self, grad_x=True, grad_cam=True, grad_phi=True, timed=False, cam_atomics=False, untiled=False,
             point_major=False):
        """Queues one fused observation pass at the current parameters.  Leaves un-normalised
        gradients in gX / red and posts the energy sum for `wait_energy()`."""
        flags = (_lib.EVAL_GRAD_X if grad_x else 0) | (_lib.EVAL_GRAD_CAM if grad_cam else 0) | \
                (_lib.EVAL_GRAD_PHI if grad_phi else 0) | (_lib.EVAL_TIMED if timed else 0) | \
                (_lib.EVAL_CAM_ATOMICS if cam_atomics else 0) | (_lib.EVAL_UNTILED if untiled else 0) | \
                (_lib.EVAL_POINT_MAJOR if point_major else 0) | \
                (_lib.EVAL_REUSE_TABLE if self._table_cam is self.cam else 0) | \
                (0 if self.distributed else _lib.EVAL_POST)
        c = self.cam.data_ptr()
        K24 = 24 * self.K
        check(self.lib.ba_eval(self.handle, self.X.data_ptr(), c, c + K24, c + 2 * K24, flags, *self._out_ptrs,
                               self.stream))
        self._table_cam = self.cam
        if self.distributed:
            self._sum_over_ranks(self.red)
            check(self.lib.ba_post_energy(self.handle, self._out_ptrs[0], self.stream))

    def wait_energy(self):
        """E = sum r^2 / (2|V|) (p2w:612) of the latest `eval`, picked up from the host mailbox the
        GPU posts it to.  Does not wait for work queued after that pass."""
        check(self.lib.ba_wait_energy(self.handle, C.byref(self._E_host), self.stream))
        if self._node_exchange:
            self._node_exchange.check()
        return self._E_host.value / (2 * self.nV)

    def arm_eval_update(self, rule, lr, bc1, bc2, X_in, X_gauge, cam_in, mom_x_in, outs, grad_cam=True, grad_phi=True,
                        timed=False):
        """Prepares the NEXT fused pass (eval_update with defer_finalize) on tensors that are not current yet - the
        results of the step queued last - into the given output tensors (X_out, s_out, r_out, X_out_alt: the points go
        to X_out if the awaited energy is an improvement, to X_out_alt (may be None) if it is not); nothing is
        launched.  wait_energy_fire launches it the moment the awaited energy allows it."""
        flags = (_lib.EVAL_GRAD_CAM if grad_cam else 0) | (_lib.EVAL_GRAD_PHI if grad_phi else 0) | \
                (_lib.EVAL_TIMED if timed else 0) | (_lib.EVAL_REUSE_TABLE if self._table_cam is cam_in else 0) | \
                _lib.EVAL_DEFER_FINALIZE
        if self.distributed:
            self._shared_sum_buffer()
        X_out, s_out, r_out, X_alt = outs
        if mom_x_in is not None:
            mp = (mom_x_in[0].data_ptr(), s_out.data_ptr(), mom_x_in[1].data_ptr(), r_out.data_ptr())
        else:
            mp = (None, None, None, None)
        c = cam_in.data_ptr()
        cam_rec = getattr(cam_in, '_ba_gauge', None)
        K24 = 24 * self.K
        o = self._out_ptrs
        check(self.lib.ba_eval_update_arm(self.handle, X_in.data_ptr(), None if X_gauge is None else X_gauge.data_ptr(),
                                          c, None if cam_rec is None else cam_rec.data_ptr(), c + K24, c + 2 * K24, flags, o[0], None, o[2], o[3], o[4], rule, float(lr),
                                          1.0 / self.nV, float(bc1), float(bc2), *mp, X_out.data_ptr(),
                                          None if X_alt is None else X_alt.data_ptr()))
        self._armed_cam = cam_in

    def wait_energy_fire(self, lo, hi, alt_lo):
        """wait_energy that launches the armed pass if lo <= E < hi (returns fired = 1) or, into the alternative
        buffer, if E >= hi and E >= alt_lo (fired = 2); returns (E, fired).  A fired pass is followed at once by the
        sum of its raw sums over the ranks, as eval_update(defer_finalize=True) does."""
        fired = C.c_int(0)
        check(self.lib.ba_wait_energy_fire(self.handle, C.byref(self._E_host), float(2 * self.nV), float(lo), float(hi),
                                           float(alt_lo), C.byref(fired), self.stream))
        if fired.value:
            self._table_cam = self._armed_cam
            if self.distributed:
                self._sum_over_ranks(self.raw)
        if self._node_exchange:
            self._node_exchange.check()
        return self._E_host.value / (2 * self.nV), int(fired.value)

    def energy(self):
        """The same value through an 8-byte device-to-host copy that synchronises the stream."""
        check(self.lib.ba_fetch_double(self._out_ptrs[0], C.byref(self._E_host), self.stream))
        if self._node_exchange:
            self._node_exchange.check()
        return self._E_host.value / (2 * self.nV)

    def eval_kernel_ms(self):
        buf = (C.c_float * 256)()
        n = C.c_int(0)
        check(self.lib.ba_eval_timing(self.handle, buf, 256, C.byref(n)))
        return [buf[k] for k in range(n.value)]

    def adam(self, x, g, s, r, lr, bc1, bc2, release=None):
        with torch.cuda.device(self.device):
            check(self.lib.ba_adam_step(x.numel(), ptr(x), ptr(g), ptr(s), ptr(r), float(lr), 1.0 / self.nV,
                                        float(bc1), float(bc2), ptr(release), stream_ptr()))
        self._table_cam = None

    def sgd(self, x, g, lr, release=None):
        with torch.cuda.device(self.device):
            check(self.lib.ba_sgd_step(x.numel(), ptr(x), ptr(g), float(lr), 1.0 / self.nV, ptr(release),
                                       stream_ptr()))
        self._table_cam = None

    def gauge_fix(self):
        """align_scene + normalize_scene in place on the current buffers (p2w:710-712)."""
        self._cam_buf = self.cam        # written in place below: a pending scale is applied for good first
        c = self.cam.data_ptr()
        check(self.lib.ba_gauge_fix(self.N, self.K, self.X.data_ptr(), c, c + 24 * self.K, ptr(self.R_gauge),
                                    ptr(self.gauge), self.stream))
        self._table_cam = None

    def eval_update(self, rule, lr, bc1, bc2, mom_x, grad_cam=True, grad_phi=True, timed=False, defer_finalize=False):
        """Queues the observation pass at the current parameters WITH the update of X fused in (see
        ba_eval_update): leaves the camera-side gradient in `red`, posts the energy, and returns
        (X_new, mom_x_new) - new tensors holding the updated points BEFORE the gauge fix of this
        step; they become current through adopt(X_new, cam_new, gauge record of camera_update).
        defer_finalize (all camera parameters free): the energy and the camera-side gradient are left
        to the camera_update(finalize=True) that must follow; with several GPUs the raw sums of the
        pass are added up over the ranks in between (one collective)."""
        cam = self._cam_buf
        cam_rec = getattr(cam, '_ba_gauge', None)
        if cam_rec is not None and self._table_cam is not cam:      # no table of these cameras: take the plain route
            cam, cam_rec = self.cam, None
        flags = (_lib.EVAL_GRAD_CAM if grad_cam else 0) | (_lib.EVAL_GRAD_PHI if grad_phi else 0) | \
                (_lib.EVAL_TIMED if timed else 0) | (_lib.EVAL_REUSE_TABLE if self._table_cam is cam else 0) | \
                (0 if self.distributed else _lib.EVAL_POST) | (_lib.EVAL_DEFER_FINALIZE if defer_finalize else 0)
        if self.distributed and defer_finalize:
            self._shared_sum_buffer()
        X_new = torch.empty_like(self._X_buf)
        if mom_x is not None:
            s_in, r_in = mom_x
            mom_x = (torch.empty_like(s_in), torch.empty_like(r_in))
            mp = (s_in.data_ptr(), mom_x[0].data_ptr(), r_in.data_ptr(), mom_x[1].data_ptr())
        else:
            mp = (None, None, None, None)
        c = cam.data_ptr()
        K24 = 24 * self.K
        o = self._out_ptrs
        check(self.lib.ba_eval_update(self.handle, self._X_buf.data_ptr(),
                                      None if self._X_gauge is None else self._X_gauge.data_ptr(), c,
                                      None if cam_rec is None else cam_rec.data_ptr(), c + K24,
                                      c + 2 * K24, flags, o[0], None, o[2], o[3], o[4], rule, float(lr),
                                      1.0 / self.nV, float(bc1), float(bc2), *mp, X_new.data_ptr(), self.stream))
        self._table_cam = cam
        if self.distributed and defer_finalize:
            self._sum_over_ranks(self.raw)
        elif self.distributed:
            self._sum_over_ranks(self.red)
            check(self.lib.ba_post_energy(self.handle, o[0], self.stream))
        return X_new, mom_x

    def camera_update(self, rule, lr, bc1, bc2, cam_range, mom_cam, gauge, in_place=False, finalize=False):
        """Queues the camera side of one step (update rule on the free slice of [C | Theta | phi_x],
        camera half of the gauge fix, camera table of the new parameters).  Returns
        (cam_new, mom_cam_new, gauge record or None); the record (a new 16-double tensor) is what
        the points still have to go through."""
        new = (lambda t: t) if in_place else torch.empty_like
        lo, hi = cam_range if cam_range is not None else (0, 0)
        if finalize:
            # merged form: no workgroup of it waits for another, so it writes new buffers, leaves the scale of the
            # gauge fix pending on the centres (the record hangs on the returned tensor) and collects the longest
            # baseline in word 14 of the record, which the previous step (or zeros(), the first time) left at zero
            if in_place:
                raise ValueError('the camera step that finishes a pass writes new buffers')
            cam_in = self._cam_buf
            cam_new = torch.empty_like(cam_in)
            if mom_cam is not None:
                s_in, r_in = mom_cam
                mom_cam = (torch.empty_like(s_in), torch.empty_like(r_in))
                mp = (s_in.data_ptr(), mom_cam[0].data_ptr(), r_in.data_ptr(), mom_cam[1].data_ptr())
            else:
                mp = (None, None, None, None)
            rec = self._next_rec if self._next_rec is not None else torch.zeros(16, dtype=F64, device=self.device)
            self._next_rec = torch.empty(16, dtype=F64, device=self.device)
            rec_in = getattr(cam_in, '_ba_gauge', None)
            check(self.lib.ba_camera_step_finalize(self.handle, cam_in.data_ptr(),
                                                   None if rec_in is None else rec_in.data_ptr(), cam_new.data_ptr(),
                                                   self.red.data_ptr(), 1, *mp, lo, hi, rule, float(lr),
                                                   1.0 / self.nV, float(bc1), float(bc2), self.R_gauge.data_ptr(),
                                                   rec.data_ptr(), self._next_rec.data_ptr(), self.stream))
            cam_new._ba_gauge = rec
            self._table_cam = cam_new
            return cam_new, mom_cam, rec
        if in_place:
            self._cam_buf = self.cam    # written in place below: a pending scale is applied for good first
        cam_new = new(self.cam)
        if mom_cam is not None:
            s_in, r_in = mom_cam
            mom_cam = (new(s_in), new(r_in))
            mp = (s_in.data_ptr(), mom_cam[0].data_ptr(), r_in.data_ptr(), mom_cam[1].data_ptr())
        else:
            mp = (None, None, None, None)
        rec = torch.empty(16, dtype=F64, device=self.device) if gauge else None
        check(self.lib.ba_camera_step(self.handle, self.cam.data_ptr(), cam_new.data_ptr(), self.red.data_ptr(),
                                      *mp, lo, hi, rule, float(lr), 1.0 / self.nV, float(bc1), float(bc2),
                                      int(bool(gauge)), self.R_gauge.data_ptr(),
                                      rec.data_ptr() if gauge else None, self.stream))
        self._table_cam = cam_new
        return cam_new, mom_cam, rec

    def step_update(self, rule, lr, bc1, bc2, x_free, cam_range, mom_x, mom_cam, gauge, in_place=False):
        """Queues one optimiser step for every free group plus the gauge fix, in two launches
        (p2w:704-712), reading the current parameters and the gradients of the last `eval`.
        rule: 1 SGD, 2 Adam.  mom_x = (s, r) tensors shaped like X or None; mom_cam = (s, r) tensors
        shaped like cam or None.  cam_range = (lo, hi) slice of [C | Theta | phi_x] that is free, or
        None.  Returns (X, cam, mom_x, mom_cam) AFTER the step: new tensors (the inputs are left
        untouched, so keeping them is a snapshot) unless in_place.  The results become current
        with `adopt`; dropping them instead discards the step."""
        new = (lambda t: t) if in_place else torch.empty_like
        X_cur = self.X          # applies a pending gauge record
        X_new, cam_new, rec = X_cur, self.cam, None
        if cam_range is not None or gauge:
            cam_new, mom_cam, rec = self.camera_update(rule, lr, bc1, bc2, cam_range, mom_cam, gauge, in_place)
        if x_free or gauge:
            X_new = new(X_cur)
            if x_free and mom_x is not None:
                s_in, r_in = mom_x
                mom_x = (new(s_in), new(r_in))
                mp = (s_in.data_ptr(), mom_x[0].data_ptr(), r_in.data_ptr(), mom_x[1].data_ptr())
            else:
                mp = (None, None, None, None)
            check(self.lib.ba_points_step(self.N, X_cur.data_ptr(), X_new.data_ptr(), self.gX.data_ptr(), *mp,
                                          rule if x_free else 0, float(lr), 1.0 / self.nV, float(bc1), float(bc2),
                                          rec.data_ptr() if gauge else None, self.stream))
        return X_new, cam_new, mom_x, mom_cam

    def triangulate(self, normalize=False):
        """Every point of the shard from the CURRENT cameras by linear triangulation (get_X_from_C_Theta,
        p2w:1655-1721) on the shard's own observation list; returns a device tensor (N, 3) in the caller's point
        order, NaN where the reference leaves NaN.  The parameters are not touched."""
        cam = self.cam
        c = cam.data_ptr()
        K24 = 24 * self.K
        X = torch.empty((self.N, 3), dtype=F64, device=self.device)
        P = torch.empty(12 * self.K, dtype=F64, device=self.device)
        check(self.lib.ba_problem_triangulate(self.handle, c, c + K24, float(cam[6 * self.K].item()), int(bool(normalize)),
                                              ptr(P), ptr(X), self.stream))
        return self.points_to_caller(X)

    # ---- second-order extension (no reference counterpart) ---------------------------------------
    def _lm_buffers(self):
        if not hasattr(self, 'V'):
            z = lambda *s: torch.zeros(*s, dtype=F64, device=self.device)
            n, ld = C.c_int64(0), C.c_int64(0)
            check(self.lib.ba_schur_dim(self.K, C.byref(n), C.byref(ld)))
            self.n_red, self.ld_red = int(n.value), int(ld.value)
            self.Xphi = z(self.N, 3)
            # camera-side blocks in one buffer (one cross-GPU sum): [B (21K) | bphi (6K) | hphi]
            self._bind_blocks(z(self.N, 6), z(27 * self.K + 1))
            self.S, self.rhs = z(self.ld_red, self.ld_red), z(self.ld_red)
            # block profile of S, written while S is built, and whether S holds zeros at the moment (the solver
            # clears what the factor occupied: at 20 000 cameras S has 115 GB, its band a few)
            self.S_first = torch.empty(self.ld_red // 64, dtype=torch.int32, device=self.device)
            self._S_is_zero = True
            self.dX, self.dcam = z(self.N, 3), z(6 * self.K + 1)
            self.X_trial = self.cam_trial = None
            self._spare_evaluation = None

    def _bind_blocks(self, V, jred):
        self.V, self.jred = V, jred
        self.B = jred[:21 * self.K].view(self.K, 21)
        self.bphi = jred[21 * self.K:27 * self.K].view(self.K, 6)
        self.hphi = jred[27 * self.K:]

    def swap_evaluation(self):
        """Exchanges the buffers the passes write (gradient, energy sum, J^T J blocks) with a second set.
        A trial point is evaluated into the other set: rejected, the first set still holds the evaluation
        of the point gone back to, and nothing is evaluated or copied again."""
        self._lm_buffers()
        mine = (self.gX, self.red, self.V, self.jred)
        other = self._spare_evaluation or tuple(torch.zeros_like(t) for t in mine)
        self._spare_evaluation = mine
        self._bind_gradient(other[0], other[1])
        self._bind_blocks(other[2], other[3])

    def rebuild_table(self):
        """Camera table of the current parameters without a pass (after going back to a kept point)."""
        cam = self.cam
        c = cam.data_ptr()
        K24 = 24 * self.K
        check(self.lib.ba_build_camera_table(self.handle, c, c + K24, c + 2 * K24, self.stream))
        self._table_cam = cam

    def eval_jtj(self, timed=False, point_major=False, dense_blocks=False, post=False):
        """Fused pass that also leaves the Gauss-Newton blocks V (this shard's points) and
        B, bphi, hphi (summed over the GPUs together with the gradient block)."""
        self._lm_buffers()
        cam = self.cam
        flags = (_lib.EVAL_TIMED if timed else 0) | (_lib.EVAL_REUSE_TABLE if self._table_cam is cam else 0) | \
                (_lib.EVAL_POINT_MAJOR if point_major else 0) | (_lib.EVAL_DENSE_BLOCKS if dense_blocks else 0)
        X = self.X
        c = cam.data_ptr()
        K24 = 24 * self.K
        check(self.lib.ba_eval_jtj(self.handle, X.data_ptr(), c, c + K24, c + 2 * K24, flags, *self._out_ptrs,
                                   ptr(self.V), ptr(self.B), ptr(self.bphi), ptr(self.hphi), self.stream))
        self._table_cam = cam
        if self.distributed:
            self._sum_over_ranks(self.red)
            self._sum_over_ranks(self.jred)
        if post:        # the energy also goes to the host mailbox (wait_energy picks it up without a copy or a stream wait)
            check(self.lib.ba_post_energy(self.handle, self._out_ptrs[0], self.stream))

    def lm_pivot(self):
        """Verdict of the last lm_step(wait=False), once the stream has been waited for (an energy read-back does):
        0, or 1 + index of the first non-positive pivot (the step and the trial point are void), or -1: the block
        profile of S changed (void as well; S is cleared and the next step reads the profile again)."""
        info = C.c_int(0)
        with torch.cuda.device(self.device):
            check(self.lib.ba_cholesky_queued_result(C.byref(info)))
        if info.value == -1:
            self.S.zero_()
        return int(info.value)

    def lm_step(self, lam, on_system=None, trial=False, wait=True):
        """Solve the damped normal equations (J^T J + lam diag) d = -J^T r by Schur complement and
        dense Cholesky; leaves dX and dcam = [dC | dTheta | dphi].  Returns the pivot info (0 = ok).
        wait=False (one GPU): nothing waits for the device - the profile of S is taken from the previous step, the
        back-substitution is queued whatever the factorisation finds, 0 is returned, and the verdict is read later
        with lm_pivot().
        trial: the back-substitution also leaves the trial point in new tensors X_trial = X + dX and
        cam_trial = cam + dcam (the current parameters are not touched).
        With several GPUs every rank adds its points' part of S and rhs, the sums go over the
        ranks, and every rank factorises the (identical) reduced system itself.
        on_system(S, rhs), if given, sees the reduced system before it is factorised in place."""
        info = C.c_int(0)
        X = self.X
        first = 1
        if self.distributed:
            first = 1 if torch.distributed.get_rank(self.group) == 0 else 0
        S_is_zero, self._S_is_zero = self._S_is_zero, False
        marks = getattr(self, 'lm_marks', None)     # a list: device time stamps around the three phases (benchmarks)

        def mark():
            if marks is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(torch.cuda.current_stream(self.device))
                marks.append(ev)
        mark()
        check(self.lib.ba_schur_build_tracked(self.handle, X.data_ptr(), float(lam), ptr(self.V), ptr(self.Xphi),
                                              ptr(self.gX), ptr(self.B), ptr(self.bphi), ptr(self.hphi), ptr(self.gC),
                                              ptr(self.gTheta), ptr(self.gphi), first, ptr(self.S), ptr(self.rhs),
                                              ptr(self.S_first), int(S_is_zero), self.stream))
        if self.distributed:
            # what the ranks add up is the envelope of S (the blocks from every block row's first occupied one to the
            # diagonal, by the profile all ranks agree on), not the matrix: 48 MB instead of 1.2 GB at 2 000 cameras,
            # 0.5 GB instead of 115 GB at 20 000
            self._min_over_ranks(self.S_first)
            nb = self.ld_red // 64
            if getattr(self, '_env_offsets', None) is None:
                self._env_offsets = torch.empty(nb + 1, dtype=torch.int64, device=self.device)
            blocks = C.c_int64(0)
            check(self.lib.ba_schur_envelope(self.ld_red, ptr(self.S_first), ptr(self._env_offsets), C.byref(blocks), self.stream))
            packed = torch.empty(blocks.value * 4096, dtype=F64, device=self.device)
            check(self.lib.ba_schur_envelope_copy(self.ld_red, ptr(self.S), ptr(self.S_first), ptr(self._env_offsets), blocks.value,
                                                  ptr(packed), 1, self.stream))
            self._sum_over_ranks(packed)
            check(self.lib.ba_schur_envelope_copy(self.ld_red, ptr(self.S), ptr(self.S_first), ptr(self._env_offsets), blocks.value,
                                                  ptr(packed), 0, self.stream))
            self._sum_over_ranks(self.rhs)
        mark()
        if on_system is not None:
            on_system(self.S, self.rhs)
        if not wait and not self.distributed:
            status = self.lib.ba_cholesky_solve_queued(self.n_red, self.ld_red, ptr(self.S), ptr(self.rhs), ptr(self.S_first),
                                                       1, self.stream)
        else:
            status = self.lib.ba_cholesky_solve_tracked(self.n_red, self.ld_red, ptr(self.S), ptr(self.rhs),
                                                        ptr(self.S_first), 1, C.byref(info), self.stream)
        self._S_is_zero = status in (0, -4)     # the solver ran to its end and cleared S
        mark()
        if self.distributed:
            # Every rank factorised the same system, but sums inside the factorisation are not ordered (the update
            # kernel adds parts of a tile with atomics): the ranks' solutions can differ in the last bits, and at a
            # borderline pivot even their verdicts.  The first rank's verdict and solution become everybody's, so the
            # replicated camera state stays bit-identical.
            verdict = torch.tensor([float(status), float(info.value)], dtype=F64, device=self.device)
            self._broadcast_first(verdict)
            status, info.value = int(verdict[0].item()), int(verdict[1].item())
            if status == 0:
                self._broadcast_first(self.rhs)
        if status == -4:
            return int(info.value)
        check(status)
        cam = self.cam
        self.X_trial = torch.empty_like(X) if trial else None
        self.cam_trial = torch.empty_like(cam) if trial else None
        check(self.lib.ba_backsubstitute(self.handle, X.data_ptr(), float(lam), ptr(self.V), ptr(self.Xphi),
                                         ptr(self.gX), ptr(self.rhs), ptr(self.dcam), ptr(self.dX), cam.data_ptr(),
                                         ptr(self.X_trial), ptr(self.cam_trial), self.stream))
        mark()
        return 0


def rotation_matrices(Theta, device='cuda:0'):
    """ryxz (p2w:138-140) on the device: (K, 3, 3)."""
    lib = _lib.load()
    dev = torch.device(device)
    Td = _dev_tensor(Theta, F64, dev).view(-1, 3)
    K = int(Td.shape[0])
    R = torch.empty((K, 3, 3), dtype=F64, device=dev)
    with torch.cuda.device(dev):
        check(lib.ba_rotation_matrices(K, ptr(Td), ptr(R), stream_ptr()))
    return R


def rotation_angles(R, device='cuda:0'):
    """get_rotations (p2w:254-267) on the device: (K, 3)."""
    lib = _lib.load()
    dev = torch.device(device)
    Rd = _dev_tensor(R, F64, dev).view(-1, 9)
    K = int(Rd.shape[0])
    Theta = torch.empty((K, 3), dtype=F64, device=dev)
    with torch.cuda.device(dev):
        check(lib.ba_rotation_angles(K, ptr(Rd), ptr(Theta), stream_ptr()))
    return Theta


def align_scene(X, C_, Theta=None, R=None, device='cuda:0'):
    """align_scene (p2w:270-303) on the device: (X_new, C_new, Theta_new, R_new)."""
    lib = _lib.load()
    dev = torch.device(device)
    Xd, Cd = _dev_tensor(X, F64, dev).view(-1, 3), _dev_tensor(C_, F64, dev).view(-1, 3)
    Td = None if Theta is None else _dev_tensor(Theta, F64, dev).view(-1, 3)
    Rd = None if R is None else _dev_tensor(R, F64, dev).view(-1, 9)
    N, K = int(Xd.shape[0]), int(Cd.shape[0])
    X_new, C_new = torch.empty_like(Xd), torch.empty_like(Cd)
    T_new = torch.empty((K, 3), dtype=F64, device=dev)
    R_new = torch.empty((K, 3, 3), dtype=F64, device=dev)
    work = torch.empty(12, dtype=F64, device=dev)
    with torch.cuda.device(dev):
        check(lib.ba_align_scene(N, K, ptr(Xd), ptr(Cd), ptr(Td), ptr(Rd), ptr(X_new), ptr(C_new), ptr(T_new), ptr(R_new),
                                 ptr(work), stream_ptr()))
    return X_new, C_new, T_new, R_new


def normalize_scene(X, C_, main_indexes=None, dist_scale=1.0, device='cuda:0'):
    """normalize_scene (p2w:306-333) on the device: (X_new, C_new, j1)."""
    lib = _lib.load()
    dev = torch.device(device)
    Xd, Cd = _dev_tensor(X, F64, dev).view(-1, 3), _dev_tensor(C_, F64, dev).view(-1, 3)
    N, K = int(Xd.shape[0]), int(Cd.shape[0])
    if main_indexes is None:
        if K < 2:       # the reference takes the argmax over the cameras after the first one (p2w:322)
            raise ValueError('attempt to get argmax of an empty sequence')
        j0, j1 = 0, -1
    else:
        j0, j1 = int(main_indexes[0]), int(main_indexes[1])
        for jx in (j1, j0):                  # numpy's indexing rules (p2w:325): -K .. K - 1, negative from the end
            if not -K <= jx < K:
                raise IndexError(f'index {jx} is out of bounds for axis 0 with size {K}')
        j0, j1 = j0 % K, j1 % K
    X_new, C_new = torch.empty_like(Xd), torch.empty_like(Cd)
    work = torch.empty(2, dtype=F64, device=dev)
    with torch.cuda.device(dev):
        check(lib.ba_normalize_scene(N, K, ptr(Xd), ptr(Cd), j0, j1, float(dist_scale), ptr(X_new), ptr(C_new), ptr(work),
                                     stream_ptr()))
    j_used = int(work[1].item())
    return X_new, C_new, (np.int64(j_used) if main_indexes is None else main_indexes[1])


def project(X, C_, Theta, phi_x, W, H, i=None, j=None, delete_back_points=True, delete_boundary_points=False,
            device='cuda:0'):
    """Model pixels for a list of (i, j) pairs, or for the dense N x K grid when i is None."""
    lib = _lib.load()
    dev = torch.device(device)
    Xd, Cd, Td = (_dev_tensor(a, F64, dev) for a in (X, C_, Theta))
    N, K = int(Xd.shape[0]), int(Cd.shape[0])
    if i is None:
        n, id_, jd = N * K, None, None
    else:
        id_, jd = _dev_tensor(i, torch.int32, dev), _dev_tensor(j, torch.int32, dev)
        n = int(id_.numel())
    out = torch.empty((n, 2), dtype=F64, device=dev)
    with torch.cuda.device(dev):
        check(lib.ba_project(N, K, n, ptr(id_), ptr(jd), ptr(Xd), ptr(Cd), ptr(Td), float(phi_x), float(W),
                             float(H), int(bool(delete_back_points)), int(bool(delete_boundary_points)),
                             ptr(out), stream_ptr()))
    return out


def reprojection_error(pix_xy, model, device='cuda:0'):
    """(sum of squared pixel differences, number of pairs) over pairs visible in both."""
    lib = _lib.load()
    dev = torch.device(device)
    pix = _dev_tensor(pix_xy, torch.int32, dev).view(-1, 2)
    mod = _dev_tensor(model, F64, dev).view(-1, 2)
    sums = torch.zeros(2, dtype=F64, device=dev)
    with torch.cuda.device(dev):
        check(lib.ba_reprojection_error(int(pix.shape[0]), ptr(pix), ptr(mod), ptr(sums), stream_ptr()))
    s = sums.cpu().numpy()
    return float(s[0]), int(s[1])


def _order_by_camera(oi, oj, N, K):
    """Indices that sort the observation list by camera, points ascending inside a camera (ba_argsort_by_camera)."""
    order = torch.empty(int(oi.numel()), dtype=torch.int32, device=oi.device)
    with torch.cuda.device(oi.device):
        check(_lib.load().ba_argsort_by_camera(int(oi.numel()), int(N), int(K), ptr(oi), ptr(oj), ptr(order), stream_ptr()))
    return order.to(torch.int64)


def _ranges(keys, n):
    """Offsets (int64, n + 1) of the runs of a sorted int32 key tensor with values in [0, n), and the run lengths
    (ba_ranges_of_sorted_keys)."""
    keys = keys.to(torch.int32).contiguous()
    rp = torch.empty(n + 1, dtype=torch.int64, device=keys.device)
    with torch.cuda.device(keys.device):
        check(_lib.load().ba_ranges_of_sorted_keys(int(keys.numel()), int(n), ptr(keys), ptr(rp), stream_ptr()))
    return rp, rp[1:] - rp[:-1]


def triangulate(oi, oj, uv, N, C_, Theta, phi_x, W, H, normalize=False, device='cuda:0', row_ptr=None):
    """Every point from the cameras that see it (ba_triangulate).  Observations sorted by point;
    returns a device tensor (N,3), NaN where the reference leaves NaN."""
    lib = _lib.load()
    dev = torch.device(device)
    oi = _dev_tensor(oi, torch.int32, dev)
    oj = _dev_tensor(oj, torch.int32, dev)
    uv = _dev_tensor(uv, F64, dev).view(-1, 2)
    Cd, Td = _dev_tensor(C_, F64, dev), _dev_tensor(Theta, F64, dev)
    K = int(Cd.shape[0])
    if oi.numel() > 1 and not bool((oi[1:] >= oi[:-1]).all()):
        raise _lib.BAError('observations must be sorted by point index')
    if row_ptr is None:
        row_ptr, counts = _ranges(oi, N)
    else:
        counts = row_ptr[1:] - row_ptr[:-1]
    # cameras that see at least one point observed twice or more (p2w:1671-1677): numpy's rank tolerance counts them
    sel = counts[oi.to(torch.int64)] >= 2
    K0 = int(torch.unique(oj[sel]).numel()) if bool(sel.any()) else 0
    X = torch.empty((N, 3), dtype=F64, device=dev)
    P = torch.empty(12 * K, dtype=F64, device=dev)
    with torch.cuda.device(dev):
        check(lib.ba_triangulate(N, K, ptr(row_ptr), ptr(oj), ptr(uv), ptr(Cd), ptr(Td), float(phi_x), float(W), float(H),
                                 int(bool(normalize)), max(2 * K0, 4), ptr(P), ptr(X), stream_ptr()))
    return X


def resect(oi, oj, uv, K, X, phi_x, W, H, normalize=False, device='cuda:0'):
    """Every camera from the points it sees (ba_resect); returns device tensors C (K,3), Theta (K,3)
    with NaN where the reference leaves NaN.  Raises ValueError like the reference when no camera
    sees 6 points (p2w:1742)."""
    lib = _lib.load()
    dev = torch.device(device)
    oi = _dev_tensor(oi, torch.int32, dev)
    oj = _dev_tensor(oj, torch.int32, dev)
    uv = _dev_tensor(uv, F64, dev).view(-1, 2)
    Xd = _dev_tensor(X, F64, dev)
    N = int(Xd.shape[0])
    order = _order_by_camera(oi, oj, N, K)
    oj_s = oj[order]
    col_ptr, counts = _ranges(oj_s, K)
    if not bool((counts >= 6).any()):
        raise ValueError('Expected each camera to have at least 6 visible points')
    sel = counts[oj.to(torch.int64)] >= 6
    N0 = int(torch.unique(oi[sel]).numel())                        # p2w:1743-1744
    oi_s, uv_s = oi[order].contiguous(), _take_rows(uv, order)
    C_out = torch.empty((K, 3), dtype=F64, device=dev)
    T_out = torch.empty((K, 3), dtype=F64, device=dev)
    with torch.cuda.device(dev):
        check(lib.ba_resect(N, K, ptr(col_ptr), ptr(oi_s), ptr(uv_s), ptr(Xd), float(phi_x), float(W), float(H),
                            int(bool(normalize)), max(2 * N0, 12), ptr(C_out), ptr(T_out), stream_ptr()))
    return C_out, T_out


def homography_diff_matrix(oi, oj, uv, N, K, normalize=False, W=None, H=None, min_common=10, device='cuda:0'):
    """Homography misfit of every camera pair with at least `min_common` common points
    (ba_homography_diff_matrix; get_H_diff_matrix, p2w:1945-1995): device tensor (K, K), NaN elsewhere.
    `uv` are pixel centres; the observations may come in any order."""
    lib = _lib.load()
    dev = torch.device(device)
    oi = _dev_tensor(oi, torch.int32, dev)
    oj = _dev_tensor(oj, torch.int32, dev)
    uv = _dev_tensor(uv, F64, dev).view(-1, 2)
    # The kernel leaves a pair at once when the point ranges of its two cameras do not overlap.  That only tells pairs
    # apart when points that are seen together are numbered together, so the points are first renumbered by their first
    # camera (the engine's own order; the result does not depend on the numbering).  Randomly numbered points of a video
    # with 2 000 cameras: 98 ms -> a few ms for the whole call.
    by_point = _order_by_camera(oj, oi, K, N)    # (roles swapped: by point, cameras ascending)
    oi, oj, uv = oi[by_point].contiguous(), oj[by_point].contiguous(), _take_rows(uv, by_point)
    perm = torch.empty(int(N), dtype=torch.int32, device=dev)
    new_i, new_j, new_uv = torch.empty_like(oi), torch.empty_like(oj), torch.empty_like(uv)
    identity = C.c_int(0)
    with torch.cuda.device(dev):
        check(lib.ba_order_points_by_first_camera(int(N), int(K), int(oi.numel()), ptr(oi), ptr(oj), ptr(uv), ptr(perm),
                                                  ptr(new_i), ptr(new_j), ptr(new_uv), C.byref(identity), stream_ptr()))
    if not identity.value:
        oi, oj, uv = new_i, new_j, new_uv
    order = _order_by_camera(oi, oj, N, K)       # by camera, points ascending
    oj_s = oj[order]
    col_ptr, _ = _ranges(oj_s, K)
    oi_s, uv_s = oi[order].contiguous(), _take_rows(uv, order)
    out = torch.empty((K, K), dtype=F64, device=dev)
    with torch.cuda.device(dev):
        check(lib.ba_homography_diff_matrix(K, ptr(col_ptr), ptr(oi_s), ptr(uv_s), float(W or 0.0), float(H or 0.0),
                                            int(bool(normalize)), int(min_common), ptr(out), stream_ptr()))
    return out
