"""qrkit_amd -- MI355X (gfx950) engine for QRKit's block-diagonal sparse QR hot path.

The product is the HIP shared library behind the C ABI of include/qrkit_amd.h;
this package is the thin Python host side used by the tests and the benchmark:
a ctypes binding (`_capi`) and a mirror of the reference's solver interface
(`SparseBlockDiagonal`, `BlockDiagonalSparseQR`) with the reference's method names.
"""
from . import _capi
from ._capi import (BLOCK_DIAGONAL_Q, COLPIV_HOUSEHOLDER, FULL_Q, HOUSEHOLDER, INFO_INVALID_INPUT,
                    INFO_SUCCESS, QrkError)
from .banded import BandedBlockedSparseQR
from .angular import BlockAngularSparseQR, BlockMatrix1x2, BlockedThinDenseQR, BlockedThinSparseQR, DenseColPivQR
from .qproduct import QProduct
from .solvers import BlockDiagonalSparseQR, Context, SparseBlockDiagonal

__all__ = ["_capi", "QrkError", "Context", "SparseBlockDiagonal", "BlockDiagonalSparseQR", "BlockMatrix1x2", "BlockAngularSparseQR", "DenseColPivQR", "BlockedThinDenseQR", "BlockedThinSparseQR", "QProduct", "BandedBlockedSparseQR", "FULL_Q",
           "BLOCK_DIAGONAL_Q", "COLPIV_HOUSEHOLDER", "HOUSEHOLDER", "INFO_SUCCESS", "INFO_INVALID_INPUT"]
