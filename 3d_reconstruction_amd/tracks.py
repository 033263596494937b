"""Keypoint-track files.

The reference keeps a scene as a pickled tuple `(X_pix, W, H)` with `X_pix` a dense integer array
(N, K, 2), -1 = not visible (demo_examples/moscow_ride/input.txt; visibility rule p2w:506).  A dense
array does not scale (N = 1 M, K = 2 k is 16 GB), so the format of this package is the list of
visible pairs: an .npz with `i`, `j` (int32, sorted by point then camera - the order of
numpy.nonzero on the dense array, p2w:531), `xy` (int32 pixels), and the scalars `N`, `K`, `W`, `H`.

`load_tracks` reads both.  The pickle is NOT unpickled: its opcode stream is only disassembled
(`pickletools.genops` executes nothing) and the one array rebuilt from its shape, dtype string and
raw bytes; anything but that layout is rejected."""
from __future__ import annotations

import pickletools

import numpy as np

_ALLOWED_STRINGS = {'numpy.core.multiarray', 'numpy._core.multiarray', '_reconstruct', 'numpy', 'ndarray', 'dtype',
                    'i4', 'i8', '<'}


def dense_to_coo(X_pix):
    """Visible pairs of a dense (N, K, 2) track array in numpy.nonzero order (p2w:506-508, 531)."""
    X_pix = np.asarray(X_pix)
    if X_pix.ndim != 3 or X_pix.shape[2] != 2:
        raise ValueError(f'Expected X_pix to have shape (N, K, 2), got {X_pix.shape}')
    vis = X_pix[:, :, 0] >= 0
    i, j = np.nonzero(vis)
    return i.astype(np.int32), j.astype(np.int32), X_pix[vis].astype(np.int32)


def coo_to_dense(i, j, xy, N, K):
    X_pix = np.full((N, K, 2), -1, dtype=np.int32)
    X_pix[i, j] = xy
    return X_pix


def read_reference_pickle(path):
    """`(X_pix, W, H)` from a file in the reference's format, without executing anything from it."""
    with open(path, 'rb') as fh:
        data = fh.read()
    ints, strs, payload, shape, after_payload = [], [], None, None, 0
    for op, arg, _ in pickletools.genops(data):
        if op.name in ('BININT', 'BININT1', 'BININT2', 'LONG1'):
            ints.append(int(arg))
        elif op.name in ('SHORT_BINUNICODE', 'BINUNICODE'):
            strs.append(arg)
        elif op.name in ('BINBYTES', 'BINBYTES8', 'SHORT_BINBYTES') and payload is None and len(arg) > 1:
            payload = arg                  # (the one-byte object before it is numpy's placeholder buffer)
            after_payload = len(ints)
        elif op.name == 'TUPLE3' and shape is None and len(ints) >= 3:
            shape = tuple(ints[-3:])       # first 3-tuple of ints: the array's shape
    dtypes = [s for s in strs if s in ('i4', 'i8')]
    if payload is None or shape is None or not set(strs) <= _ALLOWED_STRINGS or len(dtypes) != 1:
        raise ValueError(f'{path}: not a pickled (X_pix int array, W, H) tuple')
    dt = np.dtype('<' + dtypes[0])
    if len(shape) != 3 or shape[2] != 2 or int(np.prod(shape)) * dt.itemsize != len(payload):
        raise ValueError(f'{path}: array payload does not match its shape {shape}')
    if len(ints) < after_payload + 2:
        raise ValueError(f'{path}: frame size missing')
    X_pix = np.frombuffer(payload, dtype=dt).reshape(shape).astype(np.int32)
    W, H = ints[after_payload:after_payload + 2]
    return X_pix, int(W), int(H)


def save_tracks(path, i, j, xy, N, K, W, H):
    """Write the visible-pair list as .npz (sorted by point, then camera)."""
    i, j, xy = np.asarray(i, np.int32), np.asarray(j, np.int32), np.asarray(xy, np.int32).reshape(-1, 2)
    if not (i.size == j.size == xy.shape[0]):
        raise ValueError('i, j and xy must have one entry per observation')
    order = np.lexsort((j, i))
    np.savez_compressed(path, i=i[order], j=j[order], xy=xy[order], N=np.int64(N), K=np.int64(K), W=np.int64(W),
                        H=np.int64(H))


def load_tracks(path):
    """dict(i, j, xy, uv, N, K, W, H) from an .npz written by `save_tracks` or from a file in the
    reference's pickle format.  `uv` = pixel centres xy + 0.5 (p2w:531)."""
    with open(path, 'rb') as fh:
        magic = fh.read(4)
    if magic[:2] == b'PK':
        with np.load(path, allow_pickle=False) as z:
            i, j, xy = z['i'].astype(np.int32), z['j'].astype(np.int32), z['xy'].astype(np.int32)
            N, K, W, H = (int(z[k]) for k in ('N', 'K', 'W', 'H'))
    else:
        X_pix, W, H = read_reference_pickle(path)
        N, K = X_pix.shape[:2]
        i, j, xy = dense_to_coo(X_pix)
    if i.size and (i.min() < 0 or i.max() >= N or j.min() < 0 or j.max() >= K):
        raise ValueError(f'{path}: observation index out of range')
    if i.size > 1:
        key = i.astype(np.int64) * K + j
        if not (key[1:] > key[:-1]).all():
            raise ValueError(f'{path}: observations must be sorted by point then camera, without repeats')
    return dict(i=i, j=j, xy=xy, uv=xy.astype(np.float64) + 0.5, N=N, K=K, W=W, H=H)
