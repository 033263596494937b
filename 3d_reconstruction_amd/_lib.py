"""ctypes binding of libba_mi355x.so (declarations: include/ba_mi355x.h).

There is no fallback: if the library has not been built, importing the compute path raises.
`import torch` happens first on purpose so that the HIP runtime the library binds to is the
one PyTorch already loaded (both carry the soname libamdhip64.so.7)."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the library load, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libba_mi355x.so')

# enum ba_eval_flags
EVAL_GRAD_X, EVAL_GRAD_CAM, EVAL_GRAD_PHI, EVAL_CAM_ATOMICS, EVAL_TIMED = 1, 2, 4, 16, 32
EVAL_UNTILED = 512
EVAL_REUSE_TABLE = 1024
EVAL_POST = 2048
EVAL_POINT_MAJOR = 4096
EVAL_DEFER_FINALIZE = 8192
EVAL_DENSE_BLOCKS = 16384

_p = C.c_void_p
_i64 = C.c_int64
_dbl = C.c_double
_int = C.c_int

# name -> (restype, argtypes); every symbol include/ba_mi355x.h declares
SIGNATURES = {
    'ba_last_error': (C.c_char_p, []),
    'ba_version': (_int, []),
    'ba_device_info': (_int, [_int, C.POINTER(_int), C.POINTER(_int)]),
    'ba_visibility_count': (_int, [_p, _i64, _i64, _p, C.POINTER(_i64), _p]),
    'ba_visibility_fill': (_int, [_p, _i64, _i64, _p, _p, _p, _p, _p]),
    'ba_visibility_components': (_int, [_i64, _i64, _i64, _p, _p, _p, C.POINTER(_i64), C.POINTER(_i64), _p]),
    'ba_order_points_by_first_camera': (_int, [_i64, _i64, _i64] + [_p] * 7 + [C.POINTER(_int), _p]),
    'ba_argsort_by_camera': (_int, [_i64, _i64, _i64, _p, _p, _p, _p]),
    'ba_ranges_of_sorted_keys': (_int, [_i64, _i64, _p, _p, _p]),
    'ba_problem_create': (_int, [_i64, _i64, _i64, _p, _p, _p, _dbl, _dbl, _int, C.POINTER(_p)]),
    'ba_problem_destroy': (_int, [_p]),
    'ba_eval': (_int, [_p, _p, _p, _p, _p, _int, _p, _p, _p, _p, _p, _p]),
    'ba_eval_jtj': (_int, [_p, _p, _p, _p, _p, _int] + [_p] * 10),
    'ba_schur_dim': (_int, [_i64, C.POINTER(_i64), C.POINTER(_i64)]),
    'ba_schur_build': (_int, [_p, _p, _dbl] + [_p] * 9 + [_int] + [_p] * 3),
    'ba_cholesky_solve': (_int, [_i64, _i64, _p, _p, C.POINTER(_int), _p]),
    'ba_schur_build_tracked': (_int, [_p, _p, _dbl] + [_p] * 9 + [_int] + [_p] * 3 + [_int, _p]),
    'ba_cholesky_solve_tracked': (_int, [_i64, _i64, _p, _p, _p, _int, C.POINTER(_int), _p]),
    'ba_cholesky_solve_queued': (_int, [_i64, _i64, _p, _p, _p, _int, _p]),
    'ba_cholesky_queued_result': (_int, [C.POINTER(_int)]),
    'ba_cholesky_last_plan': (_int, [C.POINTER(_i64), _int, C.POINTER(_int)]),
    'ba_schur_envelope': (_int, [_i64, _p, _p, C.POINTER(_i64), _p]),
    'ba_schur_envelope_copy': (_int, [_i64, _p, _p, _p, _i64, _p, _int, _p]),
    'ba_backsubstitute': (_int, [_p, _p, _dbl, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    'ba_build_camera_table': (_int, [_p, _p, _p, _p, _p]),
    'ba_can_fuse_update': (_int, [_p]),
    'ba_eval_update': (_int, [_p] * 7 + [_int] + [_p] * 5 + [_int, _dbl, _dbl, _dbl, _dbl] + [_p] * 6),
    'ba_eval_update_arm': (_int, [_p] * 7 + [_int] + [_p] * 5 + [_int, _dbl, _dbl, _dbl, _dbl] + [_p] * 6),
    'ba_wait_energy_fire': (_int, [_p, C.POINTER(_dbl), _dbl, _dbl, _dbl, _dbl, C.POINTER(_int), _p]),
    'ba_gauge_points': (_int, [_i64, _p, _p, _p, _p]),
    'ba_scale_centres': (_int, [_i64, _p, _p, _p, _p]),
    'ba_fetch_double': (_int, [_p, C.POINTER(_dbl), _p]),
    'ba_post_energy': (_int, [_p, _p, _p]),
    'ba_wait_energy': (_int, [_p, C.POINTER(_dbl), _p]),
    'ba_eval_timing': (_int, [_p, C.POINTER(C.c_float), _int, C.POINTER(_int)]),
    'ba_adam_step': (_int, [_i64, _p, _p, _p, _p, _dbl, _dbl, _dbl, _dbl, _p, _p]),
    'ba_sgd_step': (_int, [_i64, _p, _p, _dbl, _dbl, _p, _p]),
    'ba_camera_step': (_int, [_p] * 8 + [_i64, _i64, _int, _dbl, _dbl, _dbl, _dbl, _int, _p, _p, _p]),
    'ba_camera_step_finalize': (_int, [_p, _p, _p, _p, _p, _int] + [_p] * 4 + [_i64, _i64, _int, _dbl, _dbl, _dbl, _dbl, _p, _p, _p, _p]),
    'ba_triangulate': (_int, [_i64, _i64, _p, _p, _p, _p, _p, _dbl, _dbl, _dbl, _int, _i64, _p, _p, _p]),
    'ba_problem_triangulate': (_int, [_p, _p, _p, _dbl, _int, _p, _p, _p]),
    'ba_resect': (_int, [_i64, _i64, _p, _p, _p, _p, _dbl, _dbl, _dbl, _int, _i64, _p, _p, _p]),
    'ba_sum_buffer_size': (_int, [_p, _p]),
    'ba_set_sum_buffer': (_int, [_p, _p, _i64]),
    'ba_points_step': (_int, [_i64] + [_p] * 7 + [_int, _dbl, _dbl, _dbl, _dbl, _p, _p]),
    'ba_gauge_fix': (_int, [_i64, _i64, _p, _p, _p, _p, _p, _p]),
    'ba_rotation_matrices': (_int, [_i64, _p, _p, _p]),
    'ba_rotation_angles': (_int, [_i64, _p, _p, _p]),
    'ba_align_scene': (_int, [_i64, _i64] + [_p] * 10),
    'ba_normalize_scene': (_int, [_i64, _i64, _p, _p, _i64, _i64, _dbl, _p, _p, _p, _p]),
    'ba_project': (_int, [_i64, _i64, _i64, _p, _p, _p, _p, _p, _dbl, _dbl, _dbl, _int, _int, _p, _p]),
    'ba_project_noalloc': (_int, [_i64, _i64, _i64, _p, _p, _p, _p, _p, _dbl, _dbl, _dbl, _int, _int, _p, _p]),
    'ba_reprojection_error': (_int, [_i64, _p, _p, _p, _p]),
    'ba_homography_diff_matrix': (_int, [_i64, _p, _p, _p, _dbl, _dbl, _int, _int, _p, _p]),
    'ba_subscenes_eval': (_int, [_i64, _i64] + [_p] * 7),
    'ba_subscenes_step': (_int, [_i64, _int, _dbl, _dbl, _dbl, _dbl] + [_p] * 8),
    'ba_subscenes_transform': (_int, [_i64, _i64] + [_p] * 6),
    'ba_exchange_create': (_int, [_int, _int, _i64, _p]),
    'ba_exchange_export': (_int, [_p, _p]),
    'ba_exchange_import': (_int, [_p, _int, _p]),
    'ba_exchange_sum': (_int, [_p, _p, _i64, _p]),
    'ba_exchange_status': (_int, [_p]),
    'ba_exchange_set_wait': (_int, [_p, _dbl]),
    'ba_exchange_destroy': (None, [_p]),
}

_lib = None


class BAError(RuntimeError):
    """A call into libba_mi355x.so returned a negative status."""


def load():
    """Load the library (once) and declare every entry point's signature."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f'{LIB_PATH} is missing: build it first (python -c "import __graft_entry__ as g; g.build()" '
            f'or make -C {os.path.dirname(LIB_PATH)}). There is no CPU fallback.')
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status != 0:
        msg = load().ba_last_error()
        raise BAError(f'libba_mi355x status {status}: {msg.decode() if msg else "?"}')


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
