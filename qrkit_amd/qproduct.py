"""matrixQ() of the solvers whose Q stays implicit: a product expression, like the reference's
SparseBlockYTY-based BandedBlockedSparseQRMatrixQReturnType (src/QRKit/BandedBlockedSparseQR.h:677-727),
BlockAngularSparseQRMatrixQReturnType (src/QRKit/BlockAngularSparseQR.h:651-701) and the thin solvers'
(src/QRKit/BlockedThinQRBase.h:335-470): operator*, transpose(), adjoint(), and assignment to a sparse matrix
(BandedBlockedSparseQR.h:741-765).  Every product is the solver's device path (applyQ / applyQt); this file
only shapes right-hand sides."""
from __future__ import annotations

import numpy as np
import torch


class QProduct:
    def __init__(self, solver, transposed: bool = False):
        self._s, self._t = solver, transposed

    def rows(self) -> int:
        return self._s.rows()

    def cols(self) -> int:
        return self._s.rows()

    def transpose(self) -> "QProduct":
        return QProduct(self._s, not self._t)

    adjoint = transpose        # real scalars
    T = property(transpose)

    def _apply(self, v):
        return self._s.applyQt(v) if self._t else self._s.applyQ(v)

    def __matmul__(self, other):
        """Q * other for a dense vector / matrix (numpy or torch, result of the same kind) or a scipy sparse matrix
        (result sparse, as the reference's sparse products: BandedBlockedSparseQR.h:529-633)."""
        if hasattr(other, "tocsc"):
            import scipy.sparse as sp
            other = other.tocsc()
            n = other.shape[1]
            out = []
            step = 256                                   # columns per device pass
            for c0 in range(0, n, step):
                dense = other[:, c0:c0 + step].toarray()
                out.append(sp.csc_matrix(np.asarray(self._apply(dense)).reshape(self.rows(), -1)))
            return sp.hstack(out, format="csc") if out else sp.csc_matrix((self.rows(), 0))
        return self._apply(other)

    __mul__ = __matmul__

    def toDense(self) -> np.ndarray:
        return np.asarray(self @ np.eye(self.rows()))

    def toSparse(self):
        """SparseMatrix Q = solver.matrixQ(): the product with the identity, kept sparse (Assignment glue, :741-765)."""
        import scipy.sparse as sp
        return self @ sp.identity(self.rows(), format="csc")
