"""ctypes access to oracle/libba_oracle.so (the C restatement in ba_oracle.c).

TEST INFRASTRUCTURE ONLY — same rule as ba_oracle.py: importable from tests/, smoke() and the
cpu_baseline leg of bench.py, never from the product package."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libba_oracle.so')
_lib = None
_d = np.ctypeslib.ndpointer(dtype=np.float64, flags='C_CONTIGUOUS')
_i = np.ctypeslib.ndpointer(dtype=np.int32, flags='C_CONTIGUOUS')


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f'{LIB_PATH} missing: run `make -C oracle`')
        lib = C.CDLL(LIB_PATH)
        lib.orc_energy_and_gradients.restype = C.c_double
        lib.orc_energy_and_gradients.argtypes = [C.c_int64] * 4 + [_i, _i, _d, _d, _d, _d, C.c_double, C.c_double,
                                                                   C.c_double, _d, _d, _d, _d]
        lib.orc_iteration.restype = C.c_double
        lib.orc_iteration.argtypes = [C.c_int64] * 3 + [_i, _i, _d, _d, _d, _d, _d, C.c_double, C.c_double,
                                                        C.c_double, C.c_int64] + [_d] * 12
        lib.orc_set_threads.restype = None
        lib.orc_set_threads.argtypes = [C.c_int]
        lib.orc_gauge_fix.restype = None
        lib.orc_gauge_fix.argtypes = [C.c_int64, C.c_int64, _d, _d, _d]
        _lib = lib
    return _lib


def set_threads(n):
    """OpenMP threads of the C restatement (1 by default: what the parity tests use)."""
    load().orc_set_threads(int(n))


def energy_and_gradients(X, C_, Theta, phi_x, W, H, i, j, uv, n_visible=None):
    lib = load()
    N, K, nV = X.shape[0], C_.shape[0], i.size
    gX, gC, gT, gp = np.zeros((N, 3)), np.zeros((K, 3)), np.zeros((K, 3)), np.zeros(1)
    E = lib.orc_energy_and_gradients(N, K, nV, nV if n_visible is None else n_visible,
                                     np.ascontiguousarray(i, dtype=np.int32), np.ascontiguousarray(j, dtype=np.int32),
                                     np.ascontiguousarray(uv, dtype=np.float64), np.ascontiguousarray(X),
                                     np.ascontiguousarray(C_), np.ascontiguousarray(Theta), float(phi_x), float(W),
                                     float(H), gX, gC, gT, gp)
    return E, gX, gC, gT, gp


class Iterator:
    """Runs full iterations (evaluate, Adam on every group, gauge fix) in C with caller-visible state."""

    def __init__(self, i, j, uv, X, C_, Theta, phi_x, W, H, lr):
        self.lib = load()
        self.i = np.ascontiguousarray(i, dtype=np.int32)
        self.j = np.ascontiguousarray(j, dtype=np.int32)
        self.uv = np.ascontiguousarray(uv, dtype=np.float64)
        self.X, self.C, self.Theta = X.copy(), C_.copy(), Theta.copy()
        self.phi = np.array([float(phi_x)])
        self.W, self.H, self.lr, self.t = float(W), float(H), float(lr), 1
        self.lib.orc_gauge_fix(self.X.shape[0], self.C.shape[0], self.X, self.C, self.Theta)
        z = np.zeros_like
        self.g = [z(self.X), z(self.C), z(self.Theta), z(self.phi)]
        self.s = [z(self.X), z(self.C), z(self.Theta), z(self.phi)]
        self.r = [z(self.X), z(self.C), z(self.Theta), z(self.phi)]

    def step(self):
        E = self.lib.orc_iteration(self.X.shape[0], self.C.shape[0], self.i.size, self.i, self.j, self.uv, self.X,
                                   self.C, self.Theta, self.phi, self.W, self.H, self.lr, self.t, *self.g,
                                   self.s[0], self.r[0], self.s[1], self.r[1], self.s[2], self.r[2], self.s[3],
                                   self.r[3])
        self.t += 1
        return E
