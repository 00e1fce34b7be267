"""Host-side mirror of QRKit::BandedBlockedSparseQR (src/QRKit/BandedBlockedSparseQR.h:122-366), generic
pattern path: compute(sparse matrix) = analyzePattern (as-banded-as-possible row ordering, band detection,
block merge) + factorize (sequential chain of dense Householder panels); matrixR(); implicit Q as the
ordered (Y, T) blocks with matrixQ() products; rowsPermutation(); solve().  Arithmetic and the structure
analysis both live in the HIP library (qrk_bb_*); this file only marshals buffers."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _capi as capi
from .solvers import Context


def analyze_host(J, suggestedBlockCols: int = 2):
    """The product's structure analysis without a device: (row_perm, blocks[n,4], hasPermutation)."""
    import scipy.sparse as sp
    M = sp.csr_matrix(J)
    M.sort_indices()
    rows, cols = M.shape
    rp = np.ascontiguousarray(M.indptr, dtype=np.int32)
    ci = np.ascontiguousarray(M.indices, dtype=np.int32)
    nb, has = C.c_int32(), C.c_int32()
    blocks = np.zeros(4 * rows, np.int32)
    perm = np.zeros(rows, np.int32)
    capi.check(capi.lib().qrk_bb_analyze_host(rows, cols, rp.ctypes.data, ci.ctypes.data, suggestedBlockCols, rows,
                                              C.byref(nb), blocks.ctypes.data, perm.ctypes.data, C.byref(has)))
    return perm, blocks[:4 * nb.value].reshape(-1, 4), bool(has.value)


def blocks_from_pattern(rows: int, cols: int, blockRows: int, blockCols: int, blockOverlap: int, suggestedBlockCols: int = 2):
    """BlockBandedMatrixInfo::fromBlockBandedPattern + mergeBlocks (SparseQRUtils.h:274-385) by the library's host logic
    (qrk_bb_blocks_from_pattern, no device needed): merged blocks [n, 4] = (idxRow, idxCol, numRows, numCols)."""
    nb = C.c_int32()
    cap = max(1, cols)
    blocks = np.zeros(4 * cap, np.int32)
    capi.check(capi.lib().qrk_bb_blocks_from_pattern(rows, cols, blockRows, blockCols, blockOverlap, suggestedBlockCols, cap,
                                                     C.byref(nb), blocks.ctypes.data))
    return blocks[:4 * nb.value].reshape(-1, 4)


class BandedBlockedSparseQR:
    """QRKit::BandedBlockedSparseQR<_MatrixType, _BlockQRSolver, _BlockOverlap, _SuggestedBlockCols>
    (BandedBlockedSparseQR.h:122-366).  fixedPattern=(blockRows, blockCols, blockOverlap) selects the reference's fixed-pattern
    analysis (a fixed-size _BlockQRSolver matrix type and _BlockOverlap != Dynamic, :398-408): identity row permutation and the
    block map of fromBlockBandedPattern; None (the default, = Dynamic) the generic analysis (:409-427)."""

    def __init__(self, suggestedBlockCols: int = 2, context: Optional[Context] = None, device: int = 0, fixedPattern=None):
        self._ctx = context or Context(device)
        self._suggested = suggestedBlockCols
        self._fixed = tuple(int(x) for x in fixedPattern) if fixedPattern is not None else None
        self._plan = C.c_void_p()
        self.m_isInitialized = False
        self.m_analysisIsok = False

    def compute(self, mat, forcePatternAlaysis: bool = False):
        """BandedBlockedSparseQR.h:170-182: the pattern analysis is cached unless forced."""
        if not self.m_analysisIsok or forcePatternAlaysis:
            self.analyzePattern(mat)
        self.factorize(mat)

    def analyzePattern(self, mat):
        import scipy.sparse as sp
        M = sp.csr_matrix(mat)
        M.sort_indices()
        self._rows, self._cols = M.shape
        self._rp = np.ascontiguousarray(M.indptr, dtype=np.int32)
        self._ci = np.ascontiguousarray(M.indices, dtype=np.int32)
        if self._plan:
            capi.lib().qrk_bb_plan_destroy(self._plan)
            self._plan = C.c_void_p()
        if self._fixed is not None:
            br, bc, ov = self._fixed
            capi.check(capi.lib().qrk_bb_plan_create_fixed(self._ctx.handle, self._rows, self._cols, self._rp.ctypes.data,
                                                           self._ci.ctypes.data, br, bc, ov, self._suggested,
                                                           C.byref(self._plan)), self._ctx.handle)
        else:
            capi.check(capi.lib().qrk_bb_plan_create(self._ctx.handle, self._rows, self._cols, self._rp.ctypes.data,
                                                     self._ci.ctypes.data, self._suggested, C.byref(self._plan)), self._ctx.handle)
        nb, has = C.c_int32(), C.c_int32()
        nr, yl, tl = C.c_int64(), C.c_int64(), C.c_int64()
        capi.check(capi.lib().qrk_bb_plan_info(self._plan, C.byref(nb), C.byref(nr), C.byref(yl), C.byref(tl), C.byref(has)),
                   self._ctx.handle)
        self._nb, self._nnz_r, self._ylen, self._tlen, self.hasPermutation = nb.value, nr.value, yl.value, tl.value, bool(has.value)
        self.blocks = np.zeros((self._nb, 4), np.int32)
        self.m_rowPerm = np.zeros(self._rows, np.int32)
        self.yty = np.zeros((self._nb, 6), np.int64)
        capi.check(capi.lib().qrk_bb_plan_blocks(self._plan, self.blocks.ctypes.data, self.m_rowPerm.ctypes.data,
                                                 self.yty.ctypes.data), self._ctx.handle)
        self.m_analysisIsok = True

    def factorize(self, mat):
        import scipy.sparse as sp
        M = sp.csr_matrix(mat)
        M.sort_indices()
        assert M.nnz == len(self._ci), "the sparsity pattern differs from the analysed one"
        dev = self._ctx.device
        vals = torch.from_numpy(np.ascontiguousarray(M.data, dtype=np.float64)).to(dev)
        self._r = torch.empty(max(self._nnz_r, 1), dtype=torch.float64, device=dev)
        self._y = torch.empty(max(self._ylen, 1), dtype=torch.float64, device=dev)
        self._t = torch.empty(max(self._tlen, 1), dtype=torch.float64, device=dev)
        self._ctx.use_current_stream()
        capi.check(capi.lib().qrk_bb_factorize(self._plan, vals.data_ptr(), M.nnz, self._r.data_ptr(), self._y.data_ptr(),
                                               self._t.data_ptr(), capi.MEM_DEVICE), self._ctx.handle)
        self.m_nonzeropivots = self._cols          # :513 "assuming all cols are nonzero"
        self.m_isInitialized = True
        self.m_info = capi.INFO_SUCCESS

    def rows(self):
        return self._rows

    def cols(self):
        return self._cols

    def rank(self):
        return self.m_nonzeropivots

    def info(self):
        return self.m_info

    def rowsPermutation(self) -> np.ndarray:
        return self.m_rowPerm

    def colsPermutation(self) -> np.ndarray:
        return np.arange(self._cols, dtype=np.int32)

    def matrixR(self):
        import scipy.sparse as sp
        cp = np.zeros(self._cols + 1, np.int32)
        ri = np.zeros(max(self._nnz_r, 1), np.int32)
        capi.check(capi.lib().qrk_bb_pattern(self._plan, cp.ctypes.data, ri.ctypes.data, capi.MEM_HOST), self._ctx.handle)
        return sp.csc_matrix((self._r[:self._nnz_r].cpu().numpy(), ri[:self._nnz_r], cp), shape=(self._rows, self._cols))

    def blockYTY(self, k: int):
        """(Y, T, rowIndex, numZeros) of block k, as the reference's m_blocksYT[k] (T is stored negated)."""
        row, nz, m, n, yo, to = (int(v) for v in self.yty[k])
        # the panel is stored factorised in place (row-major m x n): Y is its unit-lower view
        P = self._y[yo:yo + m * n].cpu().numpy().reshape(m, n)
        Y = np.tril(P, -1)
        Y[np.arange(min(m, n)), np.arange(min(m, n))] = 1.0
        T = self._t[to:to + n * n].cpu().numpy().reshape(n, n).T
        return Y, T, row, nz

    def _apply(self, v, transpose: bool):
        was_np = not isinstance(v, torch.Tensor)
        t = torch.as_tensor(np.asarray(v, dtype=np.float64)) if was_np else v
        shape = tuple(t.shape)
        assert shape[0] == self._rows
        x = t.reshape(self._rows, -1).t().contiguous().to(self._ctx.device, torch.float64)   # [nrhs, rows] = column-major
        self._ctx.use_current_stream()
        capi.check(capi.lib().qrk_bb_apply_q(self._plan, self._y.data_ptr(), self._t.data_ptr(), 1 if transpose else 0,
                                             x.data_ptr(), x.shape[0], capi.MEM_DEVICE), self._ctx.handle)
        out = x.t().reshape(shape)
        return out.cpu().numpy() if was_np else out

    def applyQt(self, v):
        """matrixQ().transpose() * v"""
        return self._apply(v, True)

    def applyQ(self, v):
        """matrixQ() * v"""
        return self._apply(v, False)

    def matrixQ(self):
        """Product expression (BandedBlockedSparseQR.h:677-727): matrixQ() @ v, .transpose() @ v, .toSparse()."""
        from .qproduct import QProduct
        return QProduct(self)

    def solve(self, B):
        """_solve_impl (:290-311): y = Q^T B (B already row-permuted by the caller, as in the tests :235);
        x = R(0:rank,0:rank)^-1 y(0:rank); identity column permutation."""
        was_np = not isinstance(B, torch.Tensor)
        t = torch.as_tensor(np.asarray(B, dtype=np.float64)) if was_np else B
        shape = tuple(t.shape)
        assert shape[0] == self._rows
        x = t.reshape(self._rows, -1).t().contiguous().to(self._ctx.device, torch.float64)   # [nrhs, rows] = column-major
        self._ctx.use_current_stream()
        capi.check(capi.lib().qrk_bb_apply_q(self._plan, self._y.data_ptr(), self._t.data_ptr(), 1, x.data_ptr(), x.shape[0],
                                             capi.MEM_DEVICE), self._ctx.handle)
        # back substitution with the banded R on the device (qrk_bb_solve_r), in place in the first cols entries
        capi.check(capi.lib().qrk_bb_solve_r(self._plan, x.data_ptr(), self._rows, x.shape[0], capi.MEM_DEVICE), self._ctx.handle)
        out = x[:, :self._cols].t().reshape((self._cols,) + shape[1:])
        return out.cpu().numpy() if was_np else out.contiguous()

    def __del__(self):
        try:
            if self._plan:
                capi.lib().qrk_bb_plan_destroy(self._plan)
        except Exception:
            pass


class BandedStripsQR:
    """BandedBlockedSparseQR::factorize / matrixQ / solve (src/QRKit/BandedBlockedSparseQR.h:463-508, :290-311) for a block-banded
    matrix handed over as dense strips (qrk_bbs_*, include/qrkit_amd.h): strip i is strip_rows x strip_cols, rows
    [i strip_rows, ..), columns [i col_step, i col_step + strip_cols).  Two stages on the device: every strip triangularised on
    all CUs, then a chain that merges the carried triangle with the strip's (the reference re-factorises the whole stacked panel
    per step).  R is the reference's up to row signs; 64-bit offsets throughout (BASELINE configs[2] does not fit int32)."""

    def __init__(self, num_strips: int, strip_rows: int, strip_cols: int, col_step: int, context: Optional[Context] = None, device: int = 0):
        self._ctx = context or Context(device)
        self.num_strips, self.strip_rows, self.strip_cols, self.col_step = int(num_strips), int(strip_rows), int(strip_cols), int(col_step)
        self._plan = C.c_void_p()
        capi.check(capi.lib().qrk_bbs_plan_create(self._ctx.handle, self.num_strips, self.strip_rows, self.strip_cols, self.col_step,
                                                  C.byref(self._plan)), self._ctx.handle)
        r, c, rl = C.c_int64(), C.c_int64(), C.c_int64()
        capi.check(capi.lib().qrk_bbs_plan_sizes(self._plan, C.byref(r), C.byref(c), C.byref(rl)))
        self._rows, self._cols, self._rlen = r.value, c.value, rl.value
        self.m_isInitialized = False

    def rows(self):
        return self._rows

    def cols(self):
        return self._cols

    def factorize(self, strips: torch.Tensor):
        """strips: device tensor of num_strips * strip_rows * strip_cols doubles, strip i column-major."""
        assert strips.dtype == torch.float64 and strips.is_cuda and strips.numel() == self.num_strips * self.strip_rows * self.strip_cols
        self._ctx.use_current_stream()
        capi.check(capi.lib().qrk_bbs_factorize(self._plan, strips.data_ptr()), self._ctx.handle)
        self.m_isInitialized = True
        return self

    compute = factorize

    def rRows(self, strip: int) -> torch.Tensor:
        """The rows of R emitted by strip i as a (solved x strip_cols) tensor: rows [i col_step, ..), columns [i col_step, ..)."""
        solved = self.col_step if strip + 1 < self.num_strips else self.strip_cols
        out = torch.empty(self.strip_cols, solved, dtype=torch.float64, device=self._ctx.device)
        capi.check(capi.lib().qrk_bbs_r_rows(self._plan, strip, out.data_ptr()), self._ctx.handle)
        return out.t()

    def matrixR_dense(self) -> np.ndarray:
        """Dense cols x cols upper triangle (small problems / tests only)."""
        R = np.zeros((self._cols, self._cols))
        for i in range(self.num_strips):
            blk = self.rRows(i).cpu().numpy()
            R[i * self.col_step:i * self.col_step + blk.shape[0], i * self.col_step:i * self.col_step + self.strip_cols] = blk
        return R

    def applyQ(self, v: torch.Tensor, transpose: bool) -> torch.Tensor:
        """v: (rows,) or (rows, nrhs) column-major device tensor; returns Q^T v or Q v in the layout of include/qrkit_amd.h
        (first cols entries: the rows of R)."""
        one = v.dim() == 1
        V = v.reshape(self._rows, -1)
        V = V.t().contiguous().t() if not V.t().is_contiguous() else V
        nrhs = V.shape[1]
        out = torch.empty_like(V)
        work = torch.empty(self._rows * nrhs, dtype=torch.float64, device=V.device)
        src = V.clone() if not transpose else V          # (the Q v direction reads its argument through a non-const pointer)
        self._ctx.use_current_stream()
        capi.check(capi.lib().qrk_bbs_apply_q(self._plan, 1 if transpose else 0, src.data_ptr(), out.data_ptr(), nrhs, work.data_ptr()),
                   self._ctx.handle)
        return out[:, 0] if one else out

    def solve(self, b: torch.Tensor) -> torch.Tensor:
        one = b.dim() == 1
        B = b.reshape(self._rows, -1)
        B = B.t().contiguous().t() if not B.t().is_contiguous() else B
        nrhs = B.shape[1]
        x = torch.empty(nrhs, self._cols, dtype=torch.float64, device=B.device).t()
        work = torch.empty(2 * self._rows * nrhs, dtype=torch.float64, device=B.device)
        self._ctx.use_current_stream()
        capi.check(capi.lib().qrk_bbs_solve(self._plan, B.data_ptr(), x.data_ptr(), nrhs, work.data_ptr()), self._ctx.handle)
        return x[:, 0] if one else x

    def __del__(self):
        try:
            if self._plan:
                capi.lib().qrk_bbs_plan_destroy(self._plan)
        except Exception:
            pass
