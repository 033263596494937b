"""ba_project_noalloc, the entry point that allocates nothing (its camera table lives in a fixed array of the
library), against ba_project through the C ABI: bit-identical outputs, two calls in a row, the second with a
smaller input, on sparse pair lists and on the dense grid; the size limit is refused with a clear message."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _both(lib, _lib, N, K, pairs, seed, dense=False):
    rng = np.random.default_rng(seed)
    scenes = importlib.import_module('3d_reconstruction_amd.scenes')
    sc = scenes.orbit_scene(N=N, K=K, track_len=min(K, 8), seed=seed)      # every point in front of some cameras
    Cc, Th, X = sc['C'], sc['Theta'], sc['X']
    t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(device=DEV, dtype=dt)
    Xd, Cd, Td = t(X), t(Cc), t(Th)
    if dense:
        n, i_d, j_d = N * K, None, None
    else:
        n = pairs
        # half of the pairs are true observations (inside the frame), half random (mostly behind or outside: NaN)
        pick = rng.integers(0, sc['i'].size, n // 2)
        i_d = t(np.concatenate([sc['i'][pick], rng.integers(0, N, n - n // 2)]), torch.int32)
        j_d = t(np.concatenate([sc['j'][pick], rng.integers(0, K, n - n // 2)]), torch.int32)
    ptr = _lib.ptr
    st = _lib.stream_ptr()
    outs = []
    for fn in (lib.ba_project, lib.ba_project_noalloc):
        out = torch.full((n, 2), 7.0, dtype=torch.float64, device=DEV)
        _lib.check(fn(N, K, n, ptr(i_d), ptr(j_d), ptr(Xd), ptr(Cd), ptr(Td), 1.2, 1920.0, 1080.0, 1, 1, ptr(out), st))
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    return outs


def test_project_noalloc_matches_project():
    _lib = importlib.import_module('3d_reconstruction_amd._lib')
    lib = _lib.load()
    torch.cuda.set_device(0)
    # a large call, then a smaller one: the table of the first call must not leak into the second
    for N, K, pairs, seed, dense in ((5000, 3000, 200_000, 1, False), (300, 40, 5000, 2, False), (64, 33, 0, 3, True),
                                     (7, 2, 0, 4, True)):
        want, got = _both(lib, _lib, N, K, pairs, seed, dense)
        assert np.isnan(want).any() and not np.isnan(want).all(), (N, K)
        assert np.array_equal(np.isnan(want), np.isnan(got))
        assert np.array_equal(np.nan_to_num(want, nan=-1.0), np.nan_to_num(got, nan=-1.0)), (N, K)
    # the largest table that fits, then one camera more
    want, got = _both(lib, _lib, 100, 16384, 20_000, 5)
    assert np.array_equal(np.nan_to_num(want, nan=-1.0), np.nan_to_num(got, nan=-1.0))
    z = torch.zeros(3 * 16385, dtype=torch.float64, device=DEV)
    out = torch.zeros(2, dtype=torch.float64, device=DEV)
    idx = torch.zeros(1, dtype=torch.int32, device=DEV)
    rc = lib.ba_project_noalloc(1, 16385, 1, _lib.ptr(idx), _lib.ptr(idx), _lib.ptr(z), _lib.ptr(z), _lib.ptr(z), 1.2,
                                1920.0, 1080.0, 1, 0, _lib.ptr(out), _lib.stream_ptr())
    assert rc == -1 and b'16384 cameras' in lib.ba_last_error()
    torch.cuda.synchronize()
    assert float(out.abs().sum()) == 0.0            # nothing was launched
    # no device memory changes hands in a call (the first call on the GPU, above, looked the array's address up)
    Xd = torch.zeros((50, 3), dtype=torch.float64, device=DEV)
    Xd[:, 2] = 5.0
    Cd = torch.zeros((9, 3), dtype=torch.float64, device=DEV)
    out = torch.zeros((450, 2), dtype=torch.float64, device=DEV)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        _lib.check(lib.ba_project_noalloc(50, 9, 450, None, None, _lib.ptr(Xd), _lib.ptr(Cd), _lib.ptr(Cd), 1.2, 1920.0,
                                          1080.0, 1, 0, _lib.ptr(out), _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before
    assert float(out[0, 0]) == 960.0 and float(out[0, 1]) == 540.0        # a point on the optical axis: the principal point
