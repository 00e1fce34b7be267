"""Host-side mirror of the reference's solver interface for the block-diagonal path.

Same names, argument meaning and error behaviour as
  QRKit::SparseBlockDiagonal      (src/QRKit/SparseBlockDiagonal.h:43-163)
  QRKit::BlockDiagonalSparseQR    (src/QRKit/BlockDiagonalSparseQR.h:37-335)
so that the parity tests read like test/test-qrkit.cpp:167-206.  All arithmetic
happens in the HIP library through the C ABI; torch only provides device
memory and the stream.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _capi as capi


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("qrkit_amd needs an MI355X (gfx950) device; there is no CPU fallback")


class Context:
    """One C-ABI handle bound to a device and to torch's current stream on it."""

    def __init__(self, device: int = 0):
        _require_gpu()
        self.device = torch.device("cuda", device)
        self._h = C.c_void_p()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        capi.check(capi.lib().qrk_create(C.byref(self._h), device, C.c_void_p(stream)))

    @property
    def handle(self):
        return self._h

    def use_current_stream(self):
        stream = torch.cuda.current_stream(self.device).cuda_stream
        capi.check(capi.lib().qrk_set_stream(self._h, C.c_void_p(stream)), self._h)

    def synchronize(self):
        capi.check(capi.lib().qrk_synchronize(self._h), self._h)

    def __del__(self):
        try:
            if self._h:
                capi.lib().qrk_destroy(self._h)
        except Exception:
            pass


class SparseBlockDiagonal:
    """Packed batch of dense column-major tiles = the reference's std::vector<BlockMatrixType>
    (SparseBlockDiagonal.h:159) laid out back to back, plus nRows/nCols (:161-162)."""

    def __init__(self, rows: int = 0, cols: int = 0):
        self.nRows, self.nCols = int(rows), int(cols)
        self.block_rows = np.zeros(0, np.int32)
        self.block_cols = np.zeros(0, np.int32)
        self.tiles = np.zeros(0, np.float64)   # host copy (may be None when only on device)
        self.tiles_dev: Optional[torch.Tensor] = None

    # -- construction ---------------------------------------------------------------------
    @classmethod
    def fromTiles(cls, block_rows, block_cols, tiles, rows: Optional[int] = None):
        """Tiles already cut out; `tiles` is the packed column-major value array (numpy or a
        float64 torch tensor on the device)."""
        self = cls()
        self.block_rows = np.ascontiguousarray(block_rows, dtype=np.int32)
        self.block_cols = np.ascontiguousarray(block_cols, dtype=np.int32)
        n = int((self.block_rows.astype(np.int64) * self.block_cols.astype(np.int64)).sum())
        if isinstance(tiles, torch.Tensor):
            assert tiles.dtype == torch.float64 and tiles.numel() == n
            self.tiles_dev, self.tiles = tiles.contiguous().view(-1), None
        else:
            self.tiles = np.ascontiguousarray(tiles, dtype=np.float64).reshape(-1)
            assert self.tiles.size == n
        self.nRows = int(self.block_rows.sum()) if rows is None else int(rows)
        self.nCols = int(self.block_cols.sum())
        return self

    def fromBlockDiagonalPattern(self, mat, blockRows: int, blockCols: int, context: "Optional[Context]" = None):
        """SparseBlockDiagonal::fromBlockDiagonalPattern (SparseBlockDiagonal.h:71-89) with the block
        map of BlockBandedMatrixInfo::fromBlockDiagonalPattern (SparseQRUtils.h:255-272):
        numBlocks = cols / blockCols tiles (i*blockRows, i*blockCols, blockRows, blockCols) cut out of
        `mat` (scipy sparse or dense ndarray).  A compressed sparse `mat` is cut on the device
        (qrk_bd_tiles_from_sparse) and the tiles stay there; the loop below serves dense input and the
        host-logic tests."""
        nrows, ncols = mat.shape
        num_blocks = ncols // blockCols
        if hasattr(mat, "indptr") and torch.cuda.is_available():
            return self._cutOnDevice(mat, num_blocks, blockRows, blockCols, context)
        dense_block = (lambda r0, c0: mat[r0:r0 + blockRows, c0:c0 + blockCols].toarray()) \
            if hasattr(mat, "toarray") else (lambda r0, c0: np.asarray(mat[r0:r0 + blockRows, c0:c0 + blockCols]))
        tiles = np.empty(num_blocks * blockRows * blockCols)
        for i in range(num_blocks):
            blk = dense_block(i * blockRows, i * blockCols)
            tiles[i * blockRows * blockCols:(i + 1) * blockRows * blockCols] = np.asarray(blk, dtype=np.float64).reshape(
                blockRows, blockCols).ravel(order="F")
        self.block_rows = np.full(num_blocks, blockRows, np.int32)
        self.block_cols = np.full(num_blocks, blockCols, np.int32)
        self.tiles, self.tiles_dev = tiles, None
        self.nRows, self.nCols = int(nrows), int(ncols)
        return self

    def _cutOnDevice(self, mat, num_blocks: int, blockRows, blockCols, context=None):
        """The tiles of a CSC/CSR matrix, cut by the HIP kernel behind qrk_bd_tiles_from_sparse.  blockRows/blockCols
        are either two ints (equal blocks) or per-block arrays; block i sits at the running sums of the sizes."""
        if mat.format not in ("csc", "csr"):
            mat = mat.tocsc()
        mat.sort_indices()
        ctx = context or Context(0)
        dev = ctx.device
        nrows, ncols = mat.shape
        lay = capi.BDLayout()
        lay.num_blocks = num_blocks
        if np.ndim(blockRows) == 0:
            br = np.full(num_blocks, blockRows, np.int32)
            bc = np.full(num_blocks, blockCols, np.int32)
            lay.block_rows, lay.block_cols = int(blockRows), int(blockCols)
            lay.rows = lay.cols = None
        else:
            br = np.ascontiguousarray(blockRows, dtype=np.int32)
            bc = np.ascontiguousarray(blockCols, dtype=np.int32)
            lay.rows = br.ctypes.data_as(C.POINTER(C.c_int32))
            lay.cols = bc.ctypes.data_as(C.POINTER(C.c_int32))
        n_tiles = int((br.astype(np.int64) * bc.astype(np.int64)).sum())
        lay.mat_rows, lay.mat_cols = int(nrows), int(bc.sum())
        plan = C.c_void_p()
        ctx.use_current_stream()
        capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), capi.FULL_Q, capi.COLPIV_HOUSEHOLDER,
                                                 C.byref(plan)), ctx.handle)
        try:
            ptr = torch.from_numpy(np.ascontiguousarray(mat.indptr, dtype=np.int32)).to(dev)
            idx = torch.from_numpy(np.ascontiguousarray(mat.indices, dtype=np.int32)).to(dev)
            vals = torch.from_numpy(np.ascontiguousarray(mat.data, dtype=np.float64)).to(dev)
            tiles = torch.empty(max(n_tiles, 1), dtype=torch.float64, device=dev)
            capi.check(capi.lib().qrk_bd_tiles_from_sparse(plan, 1 if mat.format == "csr" else 0, ptr.data_ptr(),
                                                           idx.data_ptr(), vals.data_ptr(), int(mat.nnz),
                                                           tiles.data_ptr(), capi.MEM_DEVICE), ctx.handle)
            ctx.synchronize()
        finally:
            capi.lib().qrk_bd_plan_destroy(plan)
        self.block_rows, self.block_cols = br, bc
        self.tiles, self.tiles_dev = None, tiles[:n_tiles]
        self.nRows, self.nCols = int(nrows), int(ncols)
        return self

    def fromSparseMatrix(self, mat, context: "Optional[Context]" = None):
        """SparseBlockDiagonal::fromSparseMatrix (SparseBlockDiagonal.h:95-130): as-banded-as-possible row ordering
        (SparseQROrdering.h:52-120), generic block detection with SuggestedBlockCols = 3 (:57,
        SparseQRUtils.h:186-253,308-385) -- both by the library's host analysis, qrk_bb_analyze_host -- then the
        blocks cut on the device.  Returns (self, rowPerm); rowPerm has Eigen's meaning, (P*M).row(rowPerm[i]) =
        M.row(i), and is the identity when the ordering found nothing to do.
        The reference cuts the blocks out of the UN-permuted `mat` (:128), which only gives the diagonal blocks when no
        permutation was needed; here they are cut out of the permuted matrix, the one the block map describes.
        The blocks must tile the diagonal (block i at the running sums of the sizes): anything else is not block
        diagonal and raises ValueError."""
        import scipy.sparse as sp
        from .banded import analyze_host
        M = sp.csr_matrix(mat)
        M.sort_indices()
        perm, blocks, has = analyze_host(M, 3)
        if has:
            inv = np.empty_like(perm)
            inv[perm] = np.arange(len(perm), dtype=perm.dtype)
            M = M[inv]
            M.sort_indices()
        else:
            perm = np.arange(M.shape[0], dtype=np.int32)
        br, bc = blocks[:, 2].astype(np.int32), blocks[:, 3].astype(np.int32)
        row0 = np.concatenate([[0], np.cumsum(br)[:-1]]) if len(br) else np.zeros(0, np.int64)
        col0 = np.concatenate([[0], np.cumsum(bc)[:-1]]) if len(bc) else np.zeros(0, np.int64)
        if not (np.array_equal(blocks[:, 0], row0) and np.array_equal(blocks[:, 1], col0) and int(bc.sum()) == M.shape[1]):
            raise ValueError("fromSparseMatrix: the detected blocks do not tile the diagonal (not a block-diagonal matrix)")
        self._cutOnDevice(M, len(br), br, bc, context)
        return self, perm

    # -- reference accessors ---------------------------------------------------------------
    def size(self) -> int:
        return len(self.block_rows)

    def rows(self) -> int:
        return self.nRows

    def cols(self) -> int:
        return self.nCols

    def clear(self):
        self.__init__(self.nRows, self.nCols)

    def __getitem__(self, i: int) -> np.ndarray:
        sizes = self.block_rows.astype(np.int64) * self.block_cols.astype(np.int64)
        off = int(sizes[:i].sum())
        r, c = int(self.block_rows[i]), int(self.block_cols[i])
        src = self.tiles if self.tiles is not None else self.tiles_dev.cpu().numpy()
        return src[off:off + r * c].reshape(c, r).T

    def is_uniform(self) -> bool:
        return self.size() > 0 and bool((self.block_rows == self.block_rows[0]).all()
                                        and (self.block_cols == self.block_cols[0]).all())

    def device_tiles(self, device) -> torch.Tensor:
        if self.tiles_dev is None or self.tiles_dev.device != device:
            self.tiles_dev = torch.from_numpy(self.tiles).to(device)
        return self.tiles_dev


class BlockDiagonalSparseQR:
    """QRKit::BlockDiagonalSparseQR<_BlockQRSolver,_QFormat> on the MI355X.

    blockSolver: COLPIV_HOUSEHOLDER (Eigen::ColPivHouseholderQR, the tests' choice) or HOUSEHOLDER.
    qFormat: FULL_Q (default) or BLOCK_DIAGONAL_Q (BlockDiagonalSparseQR.h:59-62).
    """

    def __init__(self, mat: Optional[SparseBlockDiagonal] = None, blockSolver: int = capi.COLPIV_HOUSEHOLDER,
                 qFormat: int = capi.FULL_Q, context: Optional[Context] = None, device: int = 0, hCoeffs: bool = True):
        self._ctx = context or Context(device)
        self._solver, self._qformat = blockSolver, qFormat
        # hCoeffs=False passes NULL for the optional tau output of qrk_bd_factorize: the reference never reads m_hcoeffs
        # (BlockDiagonalSparseQR.h:320), the C++ facade and bench.py do not ask for it, and the kernels are instantiated
        # without the store -- the parity tests run both instantiations
        self._want_hc = bool(hCoeffs)
        self._plan = C.c_void_p()
        self._layout_key = None
        self.m_isInitialized = False
        self.m_analysisIsok = False
        self.m_factorizationIsok = False
        self._q = self._r = self._perm = self._hc = None
        self._rowperm = None
        if mat is not None:
            self.compute(mat)

    # -- reference API ---------------------------------------------------------------------
    def compute(self, mat: SparseBlockDiagonal, rowPerm=None, forcePatternAlaysis: bool = False):
        """BlockDiagonalSparseQR.h:94-102."""
        self.analyzePattern(mat, rowPerm)
        self.m_isInitialized = False
        self.m_factorizationIsok = False
        self.factorize(mat)

    def analyzePattern(self, mat: SparseBlockDiagonal, rowPerm=None):
        """BlockDiagonalSparseQR.h:392-405: row permutation := identity or the given one; R sized."""
        self._rows, self._cols = mat.rows(), mat.cols()
        self._rowperm = np.arange(self._rows, dtype=np.int32) if rowPerm is None or len(rowPerm) == 0 \
            else np.ascontiguousarray(rowPerm, dtype=np.int32)
        key = (mat.block_rows.tobytes(), mat.block_cols.tobytes(), self._rows, self._cols)
        if key != self._layout_key:
            self._destroy_plan()
            lay = capi.BDLayout()
            lay.num_blocks = mat.size()
            lay.mat_rows, lay.mat_cols = self._rows, self._cols
            self._keep = (mat.block_rows.copy(), mat.block_cols.copy())
            if mat.is_uniform():
                lay.block_rows, lay.block_cols = int(mat.block_rows[0]), int(mat.block_cols[0])
                lay.rows = lay.cols = None
            else:
                lay.rows = self._keep[0].ctypes.data_as(C.POINTER(C.c_int32))
                lay.cols = self._keep[1].ctypes.data_as(C.POINTER(C.c_int32))
            capi.check(capi.lib().qrk_bd_plan_create(self._ctx.handle, C.byref(lay), self._qformat, self._solver,
                                                     C.byref(self._plan)), self._ctx.handle)
            self._layout_key = key
            t, q, r = C.c_int64(), C.c_int64(), C.c_int64()
            capi.check(capi.lib().qrk_bd_plan_sizes(self._plan, C.byref(t), C.byref(q), C.byref(r)), self._ctx.handle)
            self._tiles_len, self._nnz_q, self._nnz_r = t.value, q.value, r.value
        self.m_analysisIsok = True

    def factorize(self, mat: SparseBlockDiagonal):
        """BlockDiagonalSparseQR.h:415-547, on the device."""
        assert self.m_analysisIsok, "analyzePattern() should be called first"
        dev = self._ctx.device
        self._ctx.use_current_stream()
        tiles = mat.device_tiles(dev)
        self._q = torch.empty(max(self._nnz_q, 1), dtype=torch.float64, device=dev)
        self._r = torch.empty(max(self._nnz_r, 1), dtype=torch.float64, device=dev)
        self._perm = torch.empty(max(self._cols, 1), dtype=torch.int32, device=dev)
        self._hc = torch.empty(max(self._cols, 1), dtype=torch.float64, device=dev) if self._want_hc else None
        capi.check(capi.lib().qrk_bd_factorize(self._plan, tiles.data_ptr(), self._q.data_ptr(), self._r.data_ptr(),
                                               self._perm.data_ptr(), self._hc.data_ptr() if self._want_hc else None,
                                               capi.MEM_DEVICE),
                   self._ctx.handle)
        info, rank = C.c_int(), C.c_int64()
        capi.check(capi.lib().qrk_bd_info(self._plan, C.byref(info), C.byref(rank)), self._ctx.handle)
        self.m_info, self.m_nonzeropivots = info.value, rank.value
        if self.m_info != capi.INFO_SUCCESS:
            return   # reference: m_info = InvalidInput; return (before m_isInitialized is set)
        self.m_isInitialized = True
        self.m_factorizationIsok = True

    def rows(self) -> int:
        return self._rows

    def cols(self) -> int:
        return self._cols

    def rank(self) -> int:
        assert self.m_isInitialized, "The factorization should be called first, use compute()"
        return self.m_nonzeropivots

    def info(self) -> int:
        return self.m_info

    def colsPermutation(self) -> np.ndarray:
        """indices() of m_outputPerm_c: (A*P)(:, j) = A(:, indices[j])."""
        assert self.m_isInitialized, "Decomposition is not initialized."
        return self._perm[:self._cols].cpu().numpy()

    def colsPermutationDevice(self) -> torch.Tensor:
        """The same indices as an int32 tensor on the device (no copy)."""
        assert self.m_isInitialized, "Decomposition is not initialized."
        return self._perm[:self._cols]

    def rowsPermutation(self) -> np.ndarray:
        assert self.m_isInitialized, "Decomposition is not initialized."
        return self._rowperm

    def pattern(self):
        """(q_rowptr, q_colidx, r_colptr, r_rowidx) as int32 numpy arrays."""
        dev = self._ctx.device
        qp = torch.empty(self._rows + 1, dtype=torch.int32, device=dev)
        qi = torch.empty(max(self._nnz_q, 1), dtype=torch.int32, device=dev)
        rp = torch.empty(self._cols + 1, dtype=torch.int32, device=dev)
        ri = torch.empty(max(self._nnz_r, 1), dtype=torch.int32, device=dev)
        self._ctx.use_current_stream()
        capi.check(capi.lib().qrk_bd_pattern(self._plan, qp.data_ptr(), qi.data_ptr(), rp.data_ptr(), ri.data_ptr(),
                                             capi.MEM_DEVICE), self._ctx.handle)
        return (qp.cpu().numpy(), qi[:self._nnz_q].cpu().numpy(), rp.cpu().numpy(), ri[:self._nnz_r].cpu().numpy())

    def matrixQ(self):
        """Explicit sparse Q, RowMajor (scipy CSR), returned by value like the reference (:235-237)."""
        import scipy.sparse as sp
        assert self.m_isInitialized
        qp, qi, _, _ = self.pattern()
        return sp.csr_matrix((self._q[:self._nnz_q].cpu().numpy(), qi, qp), shape=(self._rows, self._rows))

    def matrixR(self):
        """Sparse R, ColMajor (scipy CSC), rows x cols."""
        import scipy.sparse as sp
        assert self.m_isInitialized
        _, _, rp, ri = self.pattern()
        return sp.csc_matrix((self._r[:self._nnz_r].cpu().numpy(), ri, rp), shape=(self._rows, self._cols))

    # device-resident value arrays (CSR order of Q, CSC order of R, tau)
    def qValues(self) -> torch.Tensor:
        return self._q[:self._nnz_q]

    def rValues(self) -> torch.Tensor:
        return self._r[:self._nnz_r]

    def hCoeffs(self) -> torch.Tensor:
        assert self._hc is not None, "constructed with hCoeffs=False"
        return self._hc[:self._cols]

    def applyQt(self, B):
        """matrixQ().transpose() * B (BlockDiagonalSparseQR.h:266; test-qrkit.cpp:187)."""
        assert self.m_isInitialized
        b, was_np, shape = self._rhs(B, self._rows)
        y = torch.empty_like(b)
        self._ctx.use_current_stream()
        capi.check(capi.lib().qrk_bd_apply_qt(self._plan, self._q.data_ptr(), b.data_ptr(), b.shape[0], y.data_ptr(),
                                              capi.MEM_DEVICE), self._ctx.handle)
        return self._out(y, was_np, shape, self._rows)

    def applyQ(self, B):
        """matrixQ() * B: the product with the explicit m_Q (:235-237) on the device (qrk_bd_apply_q)."""
        assert self.m_isInitialized
        b, was_np, shape = self._rhs(B, self._rows)
        y = torch.empty_like(b)
        self._ctx.use_current_stream()
        capi.check(capi.lib().qrk_bd_apply_q(self._plan, self._q.data_ptr(), b.data_ptr(), b.shape[0], y.data_ptr(),
                                             capi.MEM_DEVICE), self._ctx.handle)
        return self._out(y, was_np, shape, self._rows)

    def solve(self, B):
        """_solve_impl, BlockDiagonalSparseQR.h:257-299."""
        assert self.m_isInitialized, "The factorization should be called first, use compute()"
        b, was_np, shape = self._rhs(B, self._rows)
        x = torch.empty((b.shape[0], self._cols), dtype=torch.float64, device=self._ctx.device)
        self._ctx.use_current_stream()
        capi.check(capi.lib().qrk_bd_solve(self._plan, self._q.data_ptr(), self._r.data_ptr(), self._perm.data_ptr(),
                                           b.data_ptr(), b.shape[0], x.data_ptr(), capi.MEM_DEVICE), self._ctx.handle)
        self.m_info = capi.INFO_SUCCESS
        return self._out(x, was_np, shape, self._cols)

    def solveR(self, Y):
        """R.topLeftCorner(cols, cols).triangularView<Upper>().solve(Y) on the device (:271)."""
        assert self.m_isInitialized
        y, was_np, shape = self._rhs(Y, self._cols)
        z = torch.empty_like(y)
        self._ctx.use_current_stream()
        capi.check(capi.lib().qrk_bd_solve_r(self._plan, self._r.data_ptr(), y.data_ptr(), y.shape[0], z.data_ptr(),
                                             capi.MEM_DEVICE), self._ctx.handle)
        return self._out(z, was_np, shape, self._cols)

    # -- helpers ---------------------------------------------------------------------------
    def _rhs(self, B, n):
        was_np = not isinstance(B, torch.Tensor)
        t = torch.as_tensor(np.asarray(B, dtype=np.float64)) if was_np else B
        shape = tuple(t.shape)
        assert shape[0] == n, "SparseQR::solve() : invalid number of rows in the right hand side matrix"
        t2 = t.reshape(n, -1).t().contiguous().to(self._ctx.device, torch.float64)   # [nrhs, n] = column-major n x nrhs
        return t2, was_np, shape

    def _out(self, y, was_np, shape, n):
        res = y.t().reshape((y.shape[1],) + shape[1:]) if len(shape) > 1 else y.reshape(-1)
        return res.cpu().numpy() if was_np else res

    def _destroy_plan(self):
        if self._plan:
            capi.lib().qrk_bd_plan_destroy(self._plan)
            self._plan = C.c_void_p()

    def __del__(self):
        try:
            self._destroy_plan()
        except Exception:
            pass
