"""ctypes binding of the C ABI in include/qrkit_amd.h (libqrkit_amd.so, HIP/gfx950).

There is no fallback of any kind here: if the shared library is missing, or no
MI355X is visible when a context is created, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libqrkit_amd.so")

# enums of include/qrkit_amd.h
STATUS_OK, STATUS_INVALID_ARGUMENT, STATUS_NO_DEVICE, STATUS_HIP_ERROR = 0, 1, 2, 3
STATUS_ALLOC_FAILED, STATUS_UNSUPPORTED, STATUS_NOT_FACTORIZED = 4, 5, 6
INFO_SUCCESS, INFO_NUMERICAL_ISSUE, INFO_NO_CONVERGENCE, INFO_INVALID_INPUT = 0, 1, 2, 3
FULL_Q, BLOCK_DIAGONAL_Q = 0, 1
COLPIV_HOUSEHOLDER, HOUSEHOLDER = 0, 1
MEM_DEVICE, MEM_HOST = 0, 1

# every symbol the header declares, in header order
EXPORTS = (
    "qrk_version", "qrk_device_count", "qrk_create", "qrk_destroy", "qrk_set_stream", "qrk_synchronize",
    "qrk_last_error", "qrk_device_alloc", "qrk_device_free", "qrk_memcpy", "qrk_bd_plan_create", "qrk_bd_plan_destroy", "qrk_bd_plan_sizes", "qrk_bd_pattern",
    "qrk_bd_tiles_from_sparse", "qrk_bd_factorize", "qrk_bd_info", "qrk_bd_apply_qt", "qrk_bd_apply_q", "qrk_bd_solve", "qrk_bd_solve_r", "qrk_dense_plan_create",
    "qrk_dense_plan_destroy", "qrk_dense_factorize", "qrk_dense_plan_set_two_stage", "qrk_dense_plan_two_stage", "qrk_dense_apply_q", "qrk_bb_plan_create", "qrk_bb_plan_destroy", "qrk_bb_plan_create_fixed", "qrk_bb_blocks_from_pattern", "qrk_bb_analyze_host",
    "qrk_bb_plan_info", "qrk_bb_plan_blocks", "qrk_bb_pattern", "qrk_bb_factorize", "qrk_bb_apply_q", "qrk_bb_solve_r", "qrk_dense_solve_r", "qrk_bd_time_factorize", "qrk_bd_kernel_name",
    "qrk_memcpy_2d", "qrk_dense_gemv_sub", "qrk_gather_equal", "qrk_bcast", "qrk_tsqr_plan_create", "qrk_tsqr_plan_destroy", "qrk_tsqr_factorize", "qrk_tsqr_apply_q",
    "qrk_sparse_window_to_dense",
    "qrk_thin_sparse_factorize", "qrk_thin_destroy", "qrk_thin_info", "qrk_thin_matrix_r", "qrk_thin_apply_q", "qrk_thin_solve",
    "qrk_bbs_plan_create", "qrk_bbs_plan_destroy", "qrk_bbs_plan_sizes", "qrk_bbs_factorize", "qrk_bbs_r_rows", "qrk_bbs_apply_q", "qrk_bbs_solve",
    "qrk_shard_ranges", "qrk_gather_r", "qrk_gather_x",
)


class Shard(C.Structure):
    """qrk_shard of include/qrkit_amd.h."""
    _fields_ = [("first_block", C.c_int64), ("num_blocks", C.c_int64), ("base_row", C.c_int64), ("base_col", C.c_int64),
                ("tiles_off", C.c_int64), ("q_off", C.c_int64), ("r_off", C.c_int64)]


class QrkError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"qrkit_amd status {status}: {message}")
        self.status = status


class BDLayout(C.Structure):
    _fields_ = [("num_blocks", C.c_int64), ("block_rows", C.c_int32), ("block_cols", C.c_int32),
                ("rows", C.POINTER(C.c_int32)), ("cols", C.POINTER(C.c_int32)),
                ("mat_rows", C.c_int32), ("mat_cols", C.c_int32)]


_lib = None


def lib() -> C.CDLL:
    """Load libqrkit_amd.so (built by `make` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("QRKIT_AMD_LIB", LIB_PATH)   # diagnostic builds (tools/stamp_run.py) only
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build the HIP library first (make, or __graft_entry__.build()). "
            "qrkit_amd has no CPU fallback.")
    L = C.CDLL(path)
    vp, dp, ip = C.c_void_p, C.c_void_p, C.c_void_p  # device or host addresses travel as integers
    L.qrk_version.restype = C.c_int
    L.qrk_device_count.restype = C.c_int
    L.qrk_create.restype = C.c_int
    L.qrk_create.argtypes = [C.POINTER(vp), C.c_int, vp]
    L.qrk_destroy.restype = C.c_int
    L.qrk_destroy.argtypes = [vp]
    L.qrk_set_stream.restype = C.c_int
    L.qrk_set_stream.argtypes = [vp, vp]
    L.qrk_synchronize.restype = C.c_int
    L.qrk_synchronize.argtypes = [vp]
    L.qrk_last_error.restype = C.c_char_p
    L.qrk_last_error.argtypes = [vp]
    L.qrk_bd_plan_create.restype = C.c_int
    L.qrk_bd_plan_create.argtypes = [vp, C.POINTER(BDLayout), C.c_int, C.c_int, C.POINTER(vp)]
    L.qrk_bd_plan_destroy.restype = C.c_int
    L.qrk_bd_plan_destroy.argtypes = [vp]
    L.qrk_bd_plan_sizes.restype = C.c_int
    L.qrk_bd_plan_sizes.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.qrk_bd_pattern.restype = C.c_int
    L.qrk_bd_pattern.argtypes = [vp, ip, ip, ip, ip, C.c_int]
    L.qrk_bd_tiles_from_sparse.restype = C.c_int
    L.qrk_bd_tiles_from_sparse.argtypes = [vp, C.c_int, ip, ip, dp, C.c_int64, dp, C.c_int]
    L.qrk_bd_factorize.restype = C.c_int
    L.qrk_bd_factorize.argtypes = [vp, dp, dp, dp, ip, dp, C.c_int]
    L.qrk_bd_info.restype = C.c_int
    L.qrk_bd_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
    L.qrk_bd_apply_qt.restype = C.c_int
    L.qrk_bd_apply_qt.argtypes = [vp, dp, dp, C.c_int64, dp, C.c_int]
    L.qrk_bd_apply_q.restype = C.c_int
    L.qrk_bd_apply_q.argtypes = [vp, dp, dp, C.c_int64, dp, C.c_int]
    L.qrk_bd_solve.restype = C.c_int
    L.qrk_bd_solve.argtypes = [vp, dp, dp, ip, dp, C.c_int64, dp, C.c_int]
    L.qrk_bd_solve_r.restype = C.c_int
    L.qrk_bd_solve_r.argtypes = [vp, dp, dp, C.c_int64, dp, C.c_int]
    L.qrk_dense_plan_create.restype = C.c_int
    L.qrk_dense_plan_create.argtypes = [vp, C.c_int32, C.c_int32, C.c_int, C.POINTER(vp)]
    L.qrk_dense_plan_destroy.restype = C.c_int
    L.qrk_dense_plan_destroy.argtypes = [vp]
    L.qrk_dense_factorize.restype = C.c_int
    L.qrk_dense_factorize.argtypes = [vp, dp, C.c_int64, dp, ip, C.c_int]
    L.qrk_dense_plan_set_two_stage.restype = C.c_int
    L.qrk_dense_plan_set_two_stage.argtypes = [vp, C.c_int]
    L.qrk_dense_plan_two_stage.restype = C.c_int
    L.qrk_dense_plan_two_stage.argtypes = [vp]
    L.qrk_dense_apply_q.restype = C.c_int
    L.qrk_dense_apply_q.argtypes = [vp, dp, C.c_int64, dp, C.c_int, dp, C.c_int64, C.c_int64, C.c_int]
    L.qrk_dense_gemv_sub.restype = C.c_int
    L.qrk_dense_gemv_sub.argtypes = [vp, dp, C.c_int64, C.c_int64, C.c_int64, ip, dp, dp]
    L.qrk_thin_sparse_factorize.restype = C.c_int
    L.qrk_thin_sparse_factorize.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, ip, ip, dp, C.POINTER(C.c_void_p)]
    L.qrk_thin_destroy.restype = C.c_int
    L.qrk_thin_destroy.argtypes = [vp]
    L.qrk_thin_info.restype = C.c_int
    L.qrk_thin_info.argtypes = [vp, C.POINTER(C.c_int32), ip, ip]
    L.qrk_thin_matrix_r.restype = C.c_int
    L.qrk_thin_matrix_r.argtypes = [vp, dp, C.c_int64, C.c_int]
    L.qrk_thin_apply_q.restype = C.c_int
    L.qrk_thin_apply_q.argtypes = [vp, C.c_int, dp, C.c_int64, C.c_int64]
    L.qrk_thin_solve.restype = C.c_int
    L.qrk_thin_solve.argtypes = [vp, dp, C.c_int64, C.c_int64]
    L.qrk_bbs_plan_create.restype = C.c_int
    L.qrk_bbs_plan_create.argtypes = [vp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    L.qrk_bbs_plan_destroy.restype = C.c_int
    L.qrk_bbs_plan_destroy.argtypes = [vp]
    L.qrk_bbs_plan_sizes.restype = C.c_int
    L.qrk_bbs_plan_sizes.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.qrk_bbs_factorize.restype = C.c_int
    L.qrk_bbs_factorize.argtypes = [vp, dp]
    L.qrk_bbs_r_rows.restype = C.c_int
    L.qrk_bbs_r_rows.argtypes = [vp, C.c_int64, dp]
    L.qrk_bbs_apply_q.restype = C.c_int
    L.qrk_bbs_apply_q.argtypes = [vp, C.c_int, dp, dp, C.c_int64, dp]
    L.qrk_bbs_solve.restype = C.c_int
    L.qrk_bbs_solve.argtypes = [vp, dp, dp, C.c_int64, dp]
    L.qrk_bb_plan_create.restype = C.c_int
    L.qrk_bb_plan_create.argtypes = [vp, C.c_int32, C.c_int32, ip, ip, C.c_int32, C.POINTER(vp)]
    L.qrk_bb_plan_destroy.restype = C.c_int
    L.qrk_bb_plan_destroy.argtypes = [vp]
    L.qrk_bb_plan_create_fixed.restype = C.c_int
    L.qrk_bb_plan_create_fixed.argtypes = [vp, C.c_int32, C.c_int32, ip, ip, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.qrk_bb_blocks_from_pattern.restype = C.c_int
    L.qrk_bb_blocks_from_pattern.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                             C.POINTER(C.c_int32), ip]
    L.qrk_bb_analyze_host.restype = C.c_int
    L.qrk_bb_analyze_host.argtypes = [C.c_int32, C.c_int32, ip, ip, C.c_int32, C.c_int32, C.POINTER(C.c_int32), ip, ip,
                                      C.POINTER(C.c_int32)]
    L.qrk_bb_plan_info.restype = C.c_int
    L.qrk_bb_plan_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int32)]
    L.qrk_bb_plan_blocks.restype = C.c_int
    L.qrk_bb_plan_blocks.argtypes = [vp, ip, ip, ip]
    L.qrk_bb_pattern.restype = C.c_int
    L.qrk_bb_pattern.argtypes = [vp, ip, ip, C.c_int]
    L.qrk_bb_factorize.restype = C.c_int
    L.qrk_bb_factorize.argtypes = [vp, dp, C.c_int64, dp, dp, dp, C.c_int]
    L.qrk_bb_apply_q.restype = C.c_int
    L.qrk_bb_apply_q.argtypes = [vp, dp, dp, C.c_int, dp, C.c_int64, C.c_int]
    L.qrk_dense_solve_r.restype = C.c_int
    L.qrk_dense_solve_r.argtypes = [vp, dp, C.c_int64, dp, C.c_int64, C.c_int64, C.c_int]
    L.qrk_bb_solve_r.restype = C.c_int
    L.qrk_bb_solve_r.argtypes = [vp, dp, C.c_int64, C.c_int64, C.c_int]
    L.qrk_tsqr_plan_create.restype = C.c_int
    L.qrk_sparse_window_to_dense.restype = C.c_int
    L.qrk_sparse_window_to_dense.argtypes = [vp, C.c_int, C.c_int64, C.c_int64, ip, ip, dp, C.c_int64, C.c_int64, ip, dp, C.c_int64]
    L.qrk_tsqr_plan_create.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.qrk_tsqr_plan_destroy.restype = C.c_int
    L.qrk_tsqr_plan_destroy.argtypes = [vp]
    L.qrk_tsqr_factorize.restype = C.c_int
    L.qrk_tsqr_factorize.argtypes = [vp, dp, C.c_int64, C.c_int]
    L.qrk_tsqr_apply_q.restype = C.c_int
    L.qrk_tsqr_apply_q.argtypes = [vp, dp, C.c_int64, C.c_int, dp, C.c_int64, C.c_int64, C.c_int]
    L.qrk_bd_time_factorize.restype = C.c_int
    L.qrk_bd_time_factorize.argtypes = [vp, dp, dp, dp, ip, C.c_int, C.c_int, C.POINTER(C.c_float)]
    _lib = L
    return L


def check(status: int, handle=None):
    if status != STATUS_OK:
        msg = lib().qrk_last_error(handle)
        raise QrkError(status, msg.decode() if msg else "")
