"""Host-side mirror of the block-angular composition.

  QRKit::BlockMatrix1x2          (src/QRKit/BlockMatrix1x2.h:31-67)       -> BlockMatrix1x2
  QRKit::BlockAngularSparseQR    (src/QRKit/BlockAngularSparseQR.h:79-419) -> BlockAngularSparseQR
  its right-block solver, Eigen::ColPivHouseholderQR<MatrixXd> in the reference tests
  (test/test-qrkit.cpp:46-48)                                             -> DenseColPivQR

All arithmetic runs in the HIP library through the C ABI (qrk_bd_*, qrk_dense_*); torch holds the device
buffers and moves the J2 strips around (slicing / column permutation of the strip in makeR).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _capi as capi
from .solvers import BlockDiagonalSparseQR, Context, SparseBlockDiagonal


class BlockMatrix1x2:
    """Non-owning pair [left | right] (BlockMatrix1x2.h:31-67): left is a SparseBlockDiagonal,
    right a dense (rows x m2) column-major matrix (numpy or a float64 torch tensor)."""

    def __init__(self, left: SparseBlockDiagonal, right):
        self._left, self._right = left, right

    def leftBlock(self):
        return self._left

    def rightBlock(self):
        return self._right

    def rows(self) -> int:
        return int(self._right.shape[0])

    def cols(self) -> int:
        return self._left.cols() + int(self._right.shape[1])

def sparse_to_device_dense(context: Context, mat, row0: int = 0, nrows: Optional[int] = None, row_map=None) -> torch.Tensor:
    """Rows [row0, row0 + nrows) of a scipy sparse matrix as a dense COLUMN-major float64 tensor on the device
    (qrk_sparse_window_to_dense: the nonzeros cross PCIe, the dense copy the reference makes - BlockedThinSparseQR.h:131
    "m_R = mat" - is written on the device).  row_map (nrows integers, a permutation) sends source row row0 + i to row_map[i]."""
    import scipy.sparse as sp
    M = mat if sp.isspmatrix_csr(mat) or sp.isspmatrix_csc(mat) else sp.csc_matrix(mat)
    if not M.has_canonical_format:
        M = M.copy(); M.sum_duplicates()
    rows, cols = M.shape
    nrows = rows - row0 if nrows is None else int(nrows)
    dev = context.device
    outer = torch.from_numpy(np.ascontiguousarray(M.indptr, dtype=np.int32)).to(dev)
    inner = torch.from_numpy(np.ascontiguousarray(M.indices, dtype=np.int32)).to(dev)
    vals = torch.from_numpy(np.ascontiguousarray(M.data, dtype=np.float64)).to(dev)
    rmap = None if row_map is None else torch.as_tensor(np.asarray(row_map, dtype=np.int32)).to(dev)
    out = torch.empty(cols, max(nrows, 1), dtype=torch.float64, device=dev)[:, :nrows].t()   # column-major (nrows, cols)
    context.use_current_stream()
    capi.check(capi.lib().qrk_sparse_window_to_dense(
        context.handle, 1 if sp.isspmatrix_csr(M) else 0, rows, cols, outer.data_ptr(), inner.data_ptr(), vals.data_ptr(),
        int(row0), nrows, rmap.data_ptr() if rmap is not None else None, out.data_ptr(), max(nrows, 1)), context.handle)
    return out


class DenseColPivQR:
    """Dense Householder QR with implicit Q (Eigen::ColPivHouseholderQR / HouseholderQR interface)."""

    def __init__(self, context: Context, solver: int = capi.COLPIV_HOUSEHOLDER):
        self._ctx, self._solver = context, solver
        self._plan = C.c_void_p()
        self._shape = None

    def compute(self, A: torch.Tensor):
        """A: (rows, cols) float64 torch tensor on the device in COLUMN-major storage, i.e. A.t() contiguous.
        It is factorised in place (becomes the packed QR)."""
        rows, cols = A.shape
        assert A.dtype == torch.float64 and A.t().is_contiguous()
        if self._shape != (rows, cols):
            if self._plan:
                capi.lib().qrk_dense_plan_destroy(self._plan)
            self._plan = C.c_void_p()
            capi.check(capi.lib().qrk_dense_plan_create(self._ctx.handle, rows, cols, self._solver, C.byref(self._plan)),
                       self._ctx.handle)
            self._shape = (rows, cols)
        self._qr = A
        self._hc = torch.empty(max(min(rows, cols), 1), dtype=torch.float64, device=A.device)
        self._perm = torch.empty(max(cols, 1), dtype=torch.int32, device=A.device)
        self._ctx.use_current_stream()
        capi.check(capi.lib().qrk_dense_factorize(self._plan, A.data_ptr(), rows, self._hc.data_ptr(),
                                                  self._perm.data_ptr(), capi.MEM_DEVICE), self._ctx.handle)
        return self

    def rows(self):
        return self._shape[0]

    def cols(self):
        return self._shape[1]

    def rank(self):
        return min(self._shape)     # the composition only uses cols() of the right solver's rank on full-rank input

    def colsPermutation(self) -> torch.Tensor:
        return self._perm[:self._shape[1]]

    def matrixR(self) -> torch.Tensor:
        """Upper-triangular min(rows,cols) x cols (dense, device)."""
        k = min(self._shape)
        return torch.triu(self._qr[:k, :])

    def applyQ(self, B: torch.Tensor, transpose: bool) -> torch.Tensor:
        """B: (rows, nrhs) column-major device tensor, updated in place: B <- Q^T B or Q B."""
        rows = self._shape[0]
        assert B.shape[0] == rows and B.t().is_contiguous()
        self._ctx.use_current_stream()
        capi.check(capi.lib().qrk_dense_apply_q(self._plan, self._qr.data_ptr(), rows, self._hc.data_ptr(),
                                                1 if transpose else 0, B.data_ptr(), rows, B.shape[1], capi.MEM_DEVICE),
                   self._ctx.handle)
        return B

    def solveR(self, B: torch.Tensor) -> torch.Tensor:
        """B: (cols, nrhs) column-major device tensor, updated in place: B <- R^-1 B with the upper triangle of the
        packed QR (qrk_dense_solve_r; the column permutation is the caller's)."""
        rows, cols = self._shape
        assert B.shape[0] == cols and B.t().is_contiguous()
        self._ctx.use_current_stream()
        capi.check(capi.lib().qrk_dense_solve_r(self._plan, self._qr.data_ptr(), rows, B.data_ptr(), cols, B.shape[1],
                                                capi.MEM_DEVICE), self._ctx.handle)
        return B

    def __del__(self):
        try:
            if self._plan:
                capi.lib().qrk_dense_plan_destroy(self._plan)
        except Exception:
            pass


class DenseTSQR:
    """Un-pivoted communication-avoiding QR of a tall dense block on this GPU (qrk_tsqr_*): the per-rank stage of the sharded
    right solver (BlockAngularSparseQR.h:361-369 with the rows of J2 spread over GPUs)."""

    def __init__(self, context: Context):
        self._ctx = context
        self._plan = C.c_void_p()
        self._shape = None

    def compute(self, A: torch.Tensor):
        """A: (rows, cols), rows >= cols, float64, column-major on the device; factorised in place (R0 in the upper triangle of the
        first cols rows)."""
        rows, cols = A.shape
        assert A.dtype == torch.float64 and A.t().is_contiguous() and rows >= cols
        if self._shape != (rows, cols):
            if self._plan:
                capi.lib().qrk_tsqr_plan_destroy(self._plan)
            self._plan = C.c_void_p()
            capi.check(capi.lib().qrk_tsqr_plan_create(self._ctx.handle, rows, cols, C.byref(self._plan)), self._ctx.handle)
            self._shape = (rows, cols)
        self._qr = A
        self._ctx.use_current_stream()
        capi.check(capi.lib().qrk_tsqr_factorize(self._plan, A.data_ptr(), rows, capi.MEM_DEVICE), self._ctx.handle)
        return self

    def matrixR(self) -> torch.Tensor:
        return torch.triu(self._qr[:self._shape[1], :])

    def applyQ(self, B: torch.Tensor, transpose: bool) -> torch.Tensor:
        rows = self._shape[0]
        assert B.shape[0] == rows and B.t().is_contiguous()
        self._ctx.use_current_stream()
        capi.check(capi.lib().qrk_tsqr_apply_q(self._plan, self._qr.data_ptr(), rows, 1 if transpose else 0, B.data_ptr(), rows,
                                               B.shape[1], capi.MEM_DEVICE), self._ctx.handle)
        return B

    def __del__(self):
        try:
            if self._plan:
                capi.lib().qrk_tsqr_plan_destroy(self._plan)
        except Exception:
            pass


class BlockedThinDenseQR(DenseColPivQR):
    """QRKit::BlockedThinDenseQR (BlockedThinDenseQR.h:53-176): Householder QR of a thin dense matrix without column
    pivoting, Q implicit, identity permutations (:139-142).  The reference walks panels of SuggestedBlockCols columns
    (HouseholderQR of the panel at (solvedCols, solvedCols), Y/T, block-reflector update of the columns from the panel on,
    BlockedThinQRBase.h:309-333).  The reflectors of that chain ARE the reflectors of one HouseholderQR of the whole matrix
    (a panel's columns arrive updated by every earlier reflector, and inside the panel HouseholderQR is the same recurrence),
    so the device computes that factorisation (qrk_dense_*, QRK_HOUSEHOLDER) and keeps Q as essential vectors + tau instead
    of per-panel (Y, T): R and every product with Q agree with the panel chain to rounding (tests/test_thin_gpu.py compares
    with a CPU restatement of the chain); `suggestedBlockCols` therefore does not change the result."""

    def __init__(self, context: Context, suggestedBlockCols: int = 2):
        super().__init__(context, capi.HOUSEHOLDER)
        self.suggestedBlockCols = suggestedBlockCols

    def compute(self, A):
        if hasattr(A, "tocsc"):
            A = sparse_to_device_dense(self._ctx, A)              # (densified on the device: the nonzeros cross PCIe)
        if not isinstance(A, torch.Tensor):
            A = _colmajor(torch.from_numpy(np.ascontiguousarray(A, dtype=np.float64)).to(self._ctx.device))
        return super().compute(A)

    def rank(self):
        return self._shape[1]            # m_nonzeroPivots = m_R.cols() (BlockedThinDenseQR.h:132)

    def colsPermutation(self) -> torch.Tensor:
        return torch.arange(self._shape[1], dtype=torch.int32, device=self._qr.device)   # identity (:139)

    def rowsPermutation(self) -> torch.Tensor:
        return torch.arange(self._shape[0], dtype=torch.int32, device=self._qr.device)   # identity (:142)

    def _applyAny(self, v, transpose: bool):
        was_np = not isinstance(v, torch.Tensor)
        t = torch.as_tensor(np.asarray(v, dtype=np.float64)) if was_np else v
        y = _colmajor(t.to(self._ctx.device, torch.float64).reshape(self._shape[0], -1).clone())
        self.applyQ(y, transpose=transpose)
        out = y if np.ndim(v) > 1 else y[:, 0]
        return out.cpu().numpy() if was_np else out

    def matrixQ(self):
        """Product expression (BlockedThinQRBase.h:335-470)."""
        from .qproduct import QProduct
        slv = self

        class _Ops:          # applyQ / applyQt with the (rows, nrhs) calling convention of the expression
            def rows(self_inner): return slv.rows()
            def applyQ(self_inner, v): return slv._applyAny(v, False)
            def applyQt(self_inner, v): return slv._applyAny(v, True)
        return QProduct(_Ops())

    def solve(self, b: torch.Tensor) -> torch.Tensor:
        """BlockedThinQRBase::_solve_impl (BlockedThinQRBase.h:223-247): x = R(0:n,0:n)^-1 (Q^T b)(0:n)."""
        rows, cols = self._shape
        y = _colmajor(b.reshape(rows, -1).clone())
        self.applyQ(y, transpose=True)
        z = _colmajor(y[:cols, :].clone())
        return self.solveR(z)


def column_density_indices(mat) -> np.ndarray:
    """SparseQROrdering::ColumnDensity (SparseQROrdering.h:21-50): columns stable-sorted by their number of nonzeros, as the
    indices of the Eigen permutation it builds (indices[original column] = sorted rank); (A * P)(:, j) = A(:, indices[j])."""
    import scipy.sparse as sp
    M = sp.csc_matrix(mat)
    order = np.argsort(np.diff(M.indptr), kind="stable")
    idx = np.empty(M.shape[1], dtype=np.int32)
    idx[order] = np.arange(M.shape[1], dtype=np.int32)
    return idx


def as_banded_as_possible_indices(mat):
    """SparseQROrdering::AsBandedAsPossible (SparseQROrdering.h:52-120): rows stable-sorted by the column of their first
    nonzero (empty rows last); returns (indices, hasPermutation) with indices[original row] = new row."""
    import scipy.sparse as sp
    M = sp.csr_matrix(mat)
    M.sort_indices()
    rows, cols = M.shape
    start = np.full(rows, cols, dtype=np.int64)
    nz = np.diff(M.indptr) > 0
    start[nz] = M.indices[M.indptr[:-1][nz]]
    has = bool(np.any(start[1:] < start[:-1]))
    order = np.argsort(start, kind="stable") if has else np.arange(rows)
    idx = np.empty(rows, dtype=np.int32)
    idx[order] = np.arange(rows, dtype=np.int32)
    return idx, has


class BlockedThinSparseQR:
    """QRKit::BlockedThinSparseQR (src/QRKit/BlockedThinSparseQR.h:105-283) on the device, through the C entry both language
    mirrors share (qrk_thin_sparse_factorize, include/qrkit_amd.h): analyzePattern (:168-201: ColumnDensity column ordering,
    AsBandedAsPossible row ordering), compute (:105-165: the permuted matrix made dense on the device and factorised panel by
    panel -- a panel of SuggestedBlockCols columns takes the rows its sparsity pattern says, updateBlockInfo :203-238, is
    factorised by the column-pivoted dense solver, its reflectors are applied to the columns to the right, and its columns of R
    are the rows above the diagonal position plus the panel's upper triangle, :271-279).  colsPermutation() = ColumnDensity
    permutation * Householder column permutation with the zero-pivot columns last (:151-159, :250-256); rank() = nonzero pivots
    (Eigen's threshold on the panel's pivots)."""

    def __init__(self, context: Context, suggestedBlockCols: int = 2):
        self._ctx = context
        self.suggestedBlockCols = int(suggestedBlockCols)
        self._plan = C.c_void_p()
        self.m_isInitialized = False

    def _release(self):
        if self._plan:
            capi.lib().qrk_thin_destroy(self._plan)
            self._plan = C.c_void_p()

    def compute(self, mat):
        import scipy.sparse as sp
        M = sp.csc_matrix(mat)
        if not M.has_canonical_format:
            M = M.copy(); M.sum_duplicates()
        M.sort_indices()
        rows, cols = M.shape
        self._release()
        cp = np.ascontiguousarray(M.indptr, dtype=np.int32)
        ri = np.ascontiguousarray(M.indices, dtype=np.int32)
        vv = np.ascontiguousarray(M.data, dtype=np.float64)
        self._ctx.use_current_stream()
        capi.check(capi.lib().qrk_thin_sparse_factorize(self._ctx.handle, rows, cols, self.suggestedBlockCols, cp.ctypes.data,
                                                        ri.ctypes.data, vv.ctypes.data, C.byref(self._plan)), self._ctx.handle)
        rk = C.c_int32()
        cperm = np.empty(cols, np.int32); rperm = np.empty(rows, np.int32)
        capi.check(capi.lib().qrk_thin_info(self._plan, C.byref(rk), cperm.ctypes.data, rperm.ctypes.data), self._ctx.handle)
        dev = self._ctx.device
        self.m_outputPerm_c = torch.from_numpy(cperm).to(dev)
        self.m_rowPerm = torch.from_numpy(rperm).to(dev)
        self.m_nonzeroPivots = int(rk.value)
        self._shape = (rows, cols)
        self._R = None
        self.m_isInitialized = True
        return self

    def rows(self):
        return self._shape[0]

    def cols(self):
        return self._shape[1]

    def rank(self):
        return self.m_nonzeroPivots

    def colsPermutation(self) -> torch.Tensor:
        return self.m_outputPerm_c

    def rowsPermutation(self) -> torch.Tensor:
        return self.m_rowPerm

    def matrixR(self) -> torch.Tensor:
        """rows x cols, dense on the device (the reference keeps it sparse, column by column)."""
        if self._R is None:
            rows, cols = self._shape
            top = torch.empty(cols, cols, dtype=torch.float64, device=self._ctx.device)      # column-major cols x cols
            capi.check(capi.lib().qrk_thin_matrix_r(self._plan, top.data_ptr(), cols, capi.MEM_DEVICE), self._ctx.handle)
            R = torch.zeros(rows, cols, dtype=torch.float64, device=self._ctx.device)
            R[:cols] = top.t()
            self._R = R
        return self._R

    def _padded(self, v):
        """(2 rows, nrhs) column-major work copy of v with zero rows appended (the panels are applied with zero rows below them)."""
        rows = self._shape[0]
        t = v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v, dtype=np.float64))
        t = t.to(self._ctx.device, torch.float64).reshape(rows, -1)
        y = torch.zeros(t.shape[1], 2 * rows, dtype=torch.float64, device=self._ctx.device)
        y[:, :rows] = t.t()
        return y                                                     # y.t() is the column-major (2 rows, nrhs) matrix

    def _applyAny(self, v, transpose: bool):
        """SparseBlockYTY sequence (SparseBlockYTY.h:111-138): Q^T v = panels in order, Q v = in reverse; a panel acts on its
        row range only."""
        was_np = not isinstance(v, torch.Tensor)
        rows = self._shape[0]
        y = self._padded(v)
        self._ctx.use_current_stream()
        capi.check(capi.lib().qrk_thin_apply_q(self._plan, 1 if transpose else 0, y.data_ptr(), 2 * rows, y.shape[0]), self._ctx.handle)
        out = y[:, :rows].t()
        out = out if np.ndim(v) > 1 else out[:, 0]
        return out.cpu().numpy() if was_np else out

    def matrixQ(self):
        from .qproduct import QProduct
        slv = self

        class _Ops:
            def rows(self_inner): return slv.rows()
            def applyQ(self_inner, v): return slv._applyAny(v, False)
            def applyQt(self_inner, v): return slv._applyAny(v, True)
        return QProduct(_Ops())

    def solve(self, b):
        """BlockedThinQRBase::_solve_impl (BlockedThinQRBase.h:223-247): y = Q^T b; x(0:rank) = R(0:rank,0:rank)^-1 y(0:rank),
        the rest zero (the caller applies the permutations, as with the reference)."""
        rows, cols = self._shape
        was_np = not isinstance(b, torch.Tensor)
        y = self._padded(b)
        self._ctx.use_current_stream()
        capi.check(capi.lib().qrk_thin_solve(self._plan, y.data_ptr(), 2 * rows, y.shape[0]), self._ctx.handle)
        x = y[:, :cols].t()
        x = x if np.ndim(b) > 1 else x[:, 0]
        return x.cpu().numpy() if was_np else x.clone()

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass


def _colmajor(t: torch.Tensor) -> torch.Tensor:
    """Device tensor with column-major storage and the same logical shape."""
    return t.t().contiguous().t()


class BlockAngularSparseQR:
    """QRKit::BlockAngularSparseQR<BlockDiagonalSparseQR<...>, ColPivHouseholderQR<MatrixXd>>:
    QR of [J1 | J2] with J1 block diagonal and J2 dense (BlockAngularSparseQR.h:459-514)."""

    def __init__(self, context: Optional[Context] = None, device: int = 0,
                 leftBlockSolver: int = capi.COLPIV_HOUSEHOLDER, rightSolver: int = capi.COLPIV_HOUSEHOLDER):
        self._ctx = context or Context(device)
        self.m_leftSolver = BlockDiagonalSparseQR(blockSolver=leftBlockSolver, qFormat=capi.FULL_Q, context=self._ctx)
        self.m_rightSolver = DenseColPivQR(self._ctx, rightSolver)
        self.m_isInitialized = False

    def compute(self, mat: BlockMatrix1x2):
        self.analyzePattern(mat)
        self.factorize(mat)

    def analyzePattern(self, mat: BlockMatrix1x2):
        """:431-449: identity permutations, left block size."""
        left, right = mat.leftBlock(), mat.rightBlock()
        assert left.cols() > right.shape[1], "the left block should be the bigger one"
        assert left.rows() <= right.shape[0]
        self.m_leftRows, self.m_leftCols = left.rows(), left.cols()
        self._rows, self._cols = mat.rows(), mat.cols()

    def factorize(self, mat: BlockMatrix1x2):
        dev = self._ctx.device
        left, right = mat.leftBlock(), mat.rightBlock()
        m1, n1 = self.m_leftCols, self.m_leftRows
        m2 = int(right.shape[1])
        n2 = int(right.shape[0]) - n1
        # J1 = Q1 R1 (:472-475)
        self.m_leftSolver.compute(left)
        assert self.m_leftSolver.info() == capi.INFO_SUCCESS
        # solveRightBlock (:361-369): J2.top(n1) = Q1^T (rowPerm * J2.top(n1)); bottom n2 rows as they are;
        # rightSolver.compute(J2.bottomRows(n1 + n2 - m1)).  The left solver's row permutation is the identity.
        if hasattr(right, "tocsc"):          # sparse right block (test/test-qrkit.cpp:335): only its nonzeros cross PCIe
            J2 = sparse_to_device_dense(self._ctx, right)
        else:
            J2 = torch.as_tensor(np.asarray(right, dtype=np.float64)) if not isinstance(right, torch.Tensor) else right
            J2 = J2.to(dev, torch.float64)
        top = self.m_leftSolver.applyQt(J2[:n1, :])                    # device, (n1, m2)
        self._J2 = torch.cat([top, J2[n1:, :]], dim=0) if n2 > 0 else top
        bottom = torch.empty((m2, self._J2.shape[0] - m1), dtype=torch.float64, device=dev).t()    # column-major, one strided copy
        bottom.copy_(self._J2[m1:, :])
        self.m_rightSolver.compute(bottom)
        self._P2 = self.m_rightSolver.colsPermutation().long()
        self._m1, self._m2, self._n1, self._n2 = m1, m2, n1, n2
        # column permutation (:498-503) and rank (:510)
        p1 = torch.as_tensor(self.m_leftSolver.colsPermutation(), device=dev).long()
        self.m_outputPerm_c = torch.cat([p1, m1 + self._P2]).to(torch.int32)
        self.m_rowPerm = np.arange(self._rows, dtype=np.int32)
        self.m_nonzeropivots = self.m_leftSolver.rank() + min(bottom.shape)
        self.m_isInitialized = True
        self.m_info = capi.INFO_SUCCESS

    # -- accessors ----------------------------------------------------------------------------
    def rows(self):
        return self._rows

    def cols(self):
        return self._cols

    def rank(self):
        return self.m_nonzeropivots

    def info(self):
        return self.m_info

    def colsPermutation(self) -> np.ndarray:
        return self.m_outputPerm_c.cpu().numpy()

    def rowsPermutation(self) -> np.ndarray:
        return self.m_rowPerm

    def matrixR(self):
        """makeR (:285-308): R = [R1, (Q1^T J2)(0:m1, P2); 0, R2] as scipy CSC (rows x cols)."""
        import scipy.sparse as sp
        m1, m2 = self._m1, self._m2
        R1 = self.m_leftSolver.matrixR()[:, :]                            # (n1 x m1) CSC, nonzero in the top m1 rows
        strip = self._J2[:m1, :][:, self._P2].cpu().numpy()              # J2(r, P2(c)) for r < m1
        R2 = self.m_rightSolver.matrixR().cpu().numpy()                  # (min x m2)
        k2 = R2.shape[0]
        top = sp.hstack([R1[:m1, :], sp.csc_matrix(strip)], format="csc")
        mid = sp.hstack([sp.csc_matrix((k2, m1)), sp.csc_matrix(np.triu(R2))], format="csc")
        pad = sp.csc_matrix((self._rows - m1 - k2, m1 + m2))
        return sp.vstack([top, mid, pad], format="csc")

    def applyQt(self, v):
        """matrixQ().transpose() * v (:607-625): top n1 rows <- Q1^T v_top, then rows m1.. <- Q2^T of them."""
        was_np = not isinstance(v, torch.Tensor)
        t = torch.as_tensor(np.asarray(v, dtype=np.float64)) if was_np else v
        t = t.to(self._ctx.device, torch.float64).reshape(self._rows, -1).clone()
        n1, m1 = self._n1, self._m1
        t[:n1, :] = self.m_leftSolver.applyQt(t[:n1, :].contiguous())
        bot = _colmajor(t[m1:, :].clone())
        self.m_rightSolver.applyQ(bot, transpose=True)
        t[m1:, :] = bot
        out = t if np.ndim(v) > 1 else t[:, 0]
        return out.cpu().numpy() if was_np else out

    def applyQ(self, v):
        """matrixQ() * v (:627-645): rows m1.. <- Q2 of them, then the top n1 rows <- Q1 of them."""
        was_np = not isinstance(v, torch.Tensor)
        t = torch.as_tensor(np.asarray(v, dtype=np.float64)) if was_np else v
        t = t.to(self._ctx.device, torch.float64).reshape(self._rows, -1).clone()
        n1, m1 = self._n1, self._m1
        bot = _colmajor(t[m1:, :].clone())
        self.m_rightSolver.applyQ(bot, transpose=False)
        t[m1:, :] = bot
        t[:n1, :] = self.m_leftSolver.applyQ(t[:n1, :].contiguous())
        out = t if np.ndim(v) > 1 else t[:, 0]
        return out.cpu().numpy() if was_np else out

    def matrixQ(self):
        """Product expression (BlockAngularSparseQR.h:651-701): matrixQ() @ v, .transpose() @ v, .toSparse()."""
        from .qproduct import QProduct
        return QProduct(self)

    def solve(self, b):
        """_solve_impl (:202-227): x = P [R(0:rank,0:rank)^-1 (Q^T b)(0:rank)] (dense back substitution on the host
        side of the mirror is avoided: the triangular solve uses the block structure on the device)."""
        was_np = not isinstance(b, torch.Tensor)
        y = self.applyQt(torch.as_tensor(np.asarray(b, dtype=np.float64)) if was_np else b)
        y = y.reshape(self._rows, -1)
        m1, m2 = self._m1, self._m2
        y2 = self.m_rightSolver.solveR(_colmajor(y[m1:m1 + m2, :].clone()))       # R2^-1 y2 on the device (qrk_dense_solve_r)
        # y1 -= S(:, P2) z2 with the strip S = (Q1^T J2)(0:m1, :) as it lies on the device (qrk_dense_gemv_sub, one column at a time:
        # the kernel reads the strip once per right-hand side), not a library GEMM on a permuted copy
        rhs1 = _colmajor(y[:m1, :].clone())
        strip = self._J2[:m1, :]
        if not strip.t().is_contiguous():
            strip = _colmajor(strip)
        p2 = self._P2.to(torch.int32).contiguous()
        self._ctx.use_current_stream()
        for k in range(rhs1.shape[1]):
            capi.check(capi.lib().qrk_dense_gemv_sub(self._ctx.handle, strip.data_ptr(), strip.stride(1), m1, m2, p2.data_ptr(),
                                                     y2[:, k].contiguous().data_ptr(), rhs1[:, k].data_ptr()), self._ctx.handle)
        # R1 is block upper triangular: solve it tile by tile with the left solver's packed R
        y1 = self.m_leftSolver.solveR(rhs1)
        yy = torch.cat([y1, y2], dim=0)
        x = torch.empty_like(yy)
        x[self.m_outputPerm_c.long(), :] = yy
        x = x if np.ndim(b) > 1 else x[:, 0]
        return x.cpu().numpy() if was_np else x
