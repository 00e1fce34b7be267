"""ctypes/numpy loader for the CPU ORACLE (oracle/qrk_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package (qrkit_amd/) never imports it.
Parity status: floating-point values are "parity unpinned" by the reference (no
golden numbers there; Eigen absent) -- see oracle/qrk_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# QRK_ORACLE_LIB (bench.py's cpu_baseline children only): the -O3 -march=native timing copy of the same source, built by
# bench.py on the machine it times
_LIB_PATH = os.environ.get("QRK_ORACLE_LIB") or os.path.join(_HERE, "libqrk_oracle.so")

SUCCESS, NUMERICAL_ISSUE, NO_CONVERGENCE, INVALID_INPUT = 0, 1, 2, 3
FULL_Q, BLOCK_DIAGONAL_Q = 0, 1
COLPIV, NOPIV = 0, 1


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (no-op when up to date)."""
    if os.environ.get("QRK_ORACLE_LIB"):
        return _LIB_PATH
    src = os.path.join(_HERE, "qrk_oracle.c")
    hdr = os.path.join(_HERE, "qrk_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


class _Desc(C.Structure):
    _fields_ = [("B", C.c_int64), ("rows", C.POINTER(C.c_int32)), ("cols", C.POINTER(C.c_int32)),
                ("tile_off", C.POINTER(C.c_int64)), ("matRows", C.c_int32), ("matCols", C.c_int32),
                ("q_format", C.c_int), ("block_solver", C.c_int)]


class _BlockInfo(C.Structure):
    _fields_ = [("idxRow", C.c_int32), ("idxCol", C.c_int32), ("numRows", C.c_int32), ("numCols", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        dp, ip, lp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        _lib.orc_colpiv_qr.restype = C.c_int
        _lib.orc_colpiv_qr.argtypes = [dp, C.c_int, C.c_int, C.c_int, dp, ip, ip, dp]
        _lib.orc_householder_qr.restype = None
        _lib.orc_householder_qr.argtypes = [dp, C.c_int, C.c_int, C.c_int, dp]
        _lib.orc_form_q.restype = None
        _lib.orc_form_q.argtypes = [dp, C.c_int, C.c_int, C.c_int, dp, dp, C.c_int]
        _lib.orc_block_triangular_factor.restype = None
        _lib.orc_block_triangular_factor.argtypes = [dp, C.c_int, dp, C.c_int, C.c_int, C.c_int, dp]
        bi = C.POINTER(_BlockInfo)
        _lib.orc_from_block_diagonal_pattern.restype = C.c_int
        _lib.orc_from_block_diagonal_pattern.argtypes = [C.c_int32] * 4 + [bi, C.c_int]
        _lib.orc_merge_blocks.restype = C.c_int
        _lib.orc_merge_blocks.argtypes = [bi, C.c_int, C.c_int, C.c_int, bi, C.c_int]
        _lib.orc_from_block_banded_pattern.restype = C.c_int
        _lib.orc_from_block_banded_pattern.argtypes = [C.c_int32] * 5 + [C.c_int, bi, C.c_int]
        _lib.orc_block_info_from_csr.restype = C.c_int
        _lib.orc_block_info_from_csr.argtypes = [C.c_int32, C.c_int32, ip, ip, C.c_int, bi, C.c_int]
        _lib.orc_as_banded_as_possible.restype = C.c_int
        _lib.orc_as_banded_as_possible.argtypes = [C.c_int32, C.c_int32, ip, ip, ip]
        pd = C.POINTER(_Desc)
        _lib.orc_bd_sizes.restype = None
        _lib.orc_bd_sizes.argtypes = [pd, lp, lp]
        _lib.orc_bd_pattern.restype = None
        _lib.orc_bd_pattern.argtypes = [pd, ip, ip, ip, ip]
        _lib.orc_bd_factorize.restype = C.c_int
        _lib.orc_bd_factorize.argtypes = [pd, dp, dp, dp, ip, dp, lp]
        _lib.orc_bd_factorize_faithful.restype = C.c_int
        _lib.orc_bd_factorize_faithful.argtypes = [pd, dp, dp, dp, ip, lp]
        _lib.orc_bd_solve.restype = C.c_int
        _lib.orc_bd_solve.argtypes = [pd, dp, dp, ip, dp, C.c_int64, dp]
        _lib.orc_gen_reference_7x2.restype = None
        _lib.orc_gen_reference_7x2.argtypes = [C.c_int, dp]
        _lib.orc_gen_uniform.restype = None
        _lib.orc_gen_uniform.argtypes = [C.c_uint32, C.c_double, C.c_double, C.c_int64, dp]
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _lp(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


# ---------------------------------------------------------------- dense kernels

def colpiv_qr(A: np.ndarray):
    """Eigen ColPivHouseholderQR on one m x n matrix.  Returns (packedQR, hcoeffs, perm, nonzero_pivots)."""
    m, n = A.shape
    qr = np.asfortranarray(A, dtype=np.float64).copy(order="F")
    k = min(m, n)
    hc = np.zeros(k)
    tr = np.zeros(max(k, 1), dtype=np.int32)
    perm = np.zeros(max(n, 1), dtype=np.int32)
    mp = C.c_double(0.0)
    nz = lib().orc_colpiv_qr(_dp(qr), m, n, m, _dp(hc), _ip(tr), _ip(perm), C.byref(mp))
    return qr, hc, perm[:n], nz


def householder_qr(A: np.ndarray):
    m, n = A.shape
    qr = np.asfortranarray(A, dtype=np.float64).copy(order="F")
    hc = np.zeros(min(m, n))
    lib().orc_householder_qr(_dp(qr), m, n, m, _dp(hc))
    return qr, hc


def form_q(qr: np.ndarray, hc: np.ndarray) -> np.ndarray:
    m = qr.shape[0]
    qrf = np.asfortranarray(qr)
    Q = np.zeros((m, m), order="F")
    lib().orc_form_q(_dp(qrf), m, len(hc), m, _dp(np.ascontiguousarray(hc)), _dp(Q), m)
    return Q


def block_triangular_factor(V: np.ndarray, hc: np.ndarray) -> np.ndarray:
    m, n = V.shape
    Vf = np.asfortranarray(V)
    T = np.zeros((n, n), order="F")
    lib().orc_block_triangular_factor(_dp(T), n, _dp(Vf), m, n, m, _dp(np.ascontiguousarray(hc)))
    return T


# ------------------------------------------------------------------- block maps

def _bi_to_np(buf, n):
    return np.array([(b.idxRow, b.idxCol, b.numRows, b.numCols) for b in buf[:n]], dtype=np.int32).reshape(n, 4)


def from_block_diagonal_pattern(matRows, matCols, blockRows, blockCols):
    cap = max(matCols // blockCols, 1)
    buf = (_BlockInfo * cap)()
    n = lib().orc_from_block_diagonal_pattern(matRows, matCols, blockRows, blockCols, buf, cap)
    return _bi_to_np(buf, n)


def from_block_banded_pattern(matRows, matCols, blockRows, blockCols, overlap, suggested=2):
    cap = max(matCols // max(blockCols - overlap, 1), 1) + 1
    buf = (_BlockInfo * cap)()
    n = lib().orc_from_block_banded_pattern(matRows, matCols, blockRows, blockCols, overlap, suggested, buf, cap)
    return None if n < 0 else _bi_to_np(buf, n)


def merge_blocks(blocks: np.ndarray, maxColStep: int, suggested: int = 2):
    nin = len(blocks)
    inb = (_BlockInfo * max(nin, 1))()
    for i, (a, b, c, d) in enumerate(blocks):
        inb[i] = _BlockInfo(int(a), int(b), int(c), int(d))
    out = (_BlockInfo * max(nin, 1))()
    n = lib().orc_merge_blocks(inb, nin, maxColStep, suggested, out, max(nin, 1))
    return None if n < 0 else _bi_to_np(out, n)


def block_info_from_csr(rows, cols, rowptr, colidx, suggested=2):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    colidx = np.ascontiguousarray(colidx, dtype=np.int32)
    out = (_BlockInfo * max(rows, 1))()
    n = lib().orc_block_info_from_csr(rows, cols, _ip(rowptr), _ip(colidx), suggested, out, max(rows, 1))
    return None if n < 0 else _bi_to_np(out, n)


def as_banded_as_possible(rows, cols, rowptr, colidx):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    colidx = np.ascontiguousarray(colidx, dtype=np.int32)
    perm = np.zeros(max(rows, 1), dtype=np.int32)
    has = lib().orc_as_banded_as_possible(rows, cols, _ip(rowptr), _ip(colidx), _ip(perm))
    return bool(has), perm[:rows]


# ------------------------------------------------------- BlockDiagonalSparseQR

@dataclass
class BDResult:
    info: int
    rank: int
    Q_vals: np.ndarray
    R_vals: np.ndarray
    perm: np.ndarray
    hcoeffs: np.ndarray


class BDProblem:
    """A packed batch of column-major tiles = the reference's SparseBlockDiagonal."""

    def __init__(self, rows, cols, tiles, matRows=None, q_format=FULL_Q, block_solver=COLPIV):
        self.rows = np.ascontiguousarray(rows, dtype=np.int32)
        self.cols = np.ascontiguousarray(cols, dtype=np.int32)
        self.B = len(self.rows)
        sizes = self.rows.astype(np.int64) * self.cols.astype(np.int64)
        self.tile_off = np.zeros(self.B + 1, dtype=np.int64)
        np.cumsum(sizes, out=self.tile_off[1:])
        self.tiles = np.ascontiguousarray(tiles, dtype=np.float64).reshape(-1)
        assert self.tiles.size == self.tile_off[-1]
        self.matRows = int(self.rows.sum()) if matRows is None else int(matRows)
        self.matCols = int(self.cols.sum())
        self.q_format, self.block_solver = q_format, block_solver
        self._d = _Desc(self.B, _ip(self.rows), _ip(self.cols), _lp(self.tile_off), self.matRows,
                        self.matCols, q_format, block_solver)

    @classmethod
    def uniform(cls, B, r, c, tiles, **kw):
        return cls(np.full(B, r, np.int32), np.full(B, c, np.int32), tiles, **kw)

    def sizes(self):
        q, r = C.c_int64(0), C.c_int64(0)
        lib().orc_bd_sizes(C.byref(self._d), C.byref(q), C.byref(r))
        return q.value, r.value

    def pattern(self):
        nq, nr = self.sizes()
        q_rowptr = np.zeros(self.matRows + 1, np.int32)
        q_colidx = np.zeros(max(nq, 1), np.int32)
        r_colptr = np.zeros(self.matCols + 1, np.int32)
        r_rowidx = np.zeros(max(nr, 1), np.int32)
        lib().orc_bd_pattern(C.byref(self._d), _ip(q_rowptr), _ip(q_colidx), _ip(r_colptr), _ip(r_rowidx))
        return q_rowptr, q_colidx[:nq], r_colptr, r_rowidx[:nr]

    def factorize(self, faithful: bool = False) -> BDResult:
        nq, nr = self.sizes()
        Q = np.zeros(max(nq, 1))
        R = np.zeros(max(nr, 1))
        perm = np.zeros(max(self.matCols, 1), np.int32)
        hc = np.zeros(max(self.matCols, 1))
        rank = C.c_int64(0)
        if faithful:
            info = lib().orc_bd_factorize_faithful(C.byref(self._d), _dp(self.tiles), _dp(Q), _dp(R), _ip(perm),
                                                   C.byref(rank))
        else:
            info = lib().orc_bd_factorize(C.byref(self._d), _dp(self.tiles), _dp(Q), _dp(R), _ip(perm), _dp(hc),
                                          C.byref(rank))
        return BDResult(info, rank.value, Q[:nq], R[:nr], perm[:self.matCols], hc[:self.matCols])

    def solve(self, res: BDResult, b: np.ndarray) -> np.ndarray:
        b2 = np.asfortranarray(b.reshape(self.matRows, -1), dtype=np.float64)
        nrhs = b2.shape[1]
        x = np.zeros((self.matCols, nrhs), order="F")
        info = lib().orc_bd_solve(C.byref(self._d), _dp(res.Q_vals), _dp(res.R_vals), _ip(res.perm), _dp(b2), nrhs,
                                  _dp(x))
        assert info == SUCCESS
        return x.reshape(b.shape[:0] + (self.matCols,) + b.shape[1:]) if b.ndim > 1 else x[:, 0]


# -------------------------------------------------------------------- generators

def gen_reference_7x2(numVars: int) -> np.ndarray:
    """Tiles of generate_block_diagonal_matrix (test/test-qrkit.cpp:101-117), shape (numVars, 2, 7) = col-major 7x2."""
    t = np.zeros(numVars * 14)
    lib().orc_gen_reference_7x2(numVars, _dp(t))
    return t


def gen_uniform(seed: int, lo: float, hi: float, n: int) -> np.ndarray:
    t = np.zeros(n)
    lib().orc_gen_uniform(seed, lo, hi, n, _dp(t))
    return t


# ---------------------------------------------------------------- BlockAngularSparseQR

class BAResult:
    pass


def ba_factorize(prob: BDProblem, J2: np.ndarray, right_solver: int = COLPIV) -> BAResult:
    """BlockAngularSparseQR::factorize (src/QRKit/BlockAngularSparseQR.h:459-514) with a block-diagonal left
    solver (FullQ) and a dense Eigen right solver, restated with numpy on top of the C kernels above:
      solveRightBlock (:361-369): J2.top(n1) <- Q1^T J2.top(n1); rightSolver.compute(J2.bottomRows(n1+n2-m1));
      makeR (:285-308): R = [R1, J2(0:m1, P2); 0, R2];  perm = [P1; m1 + P2] (:498-503); rank (:510).
    """
    import scipy.sparse as sp
    assert prob.q_format == FULL_Q
    res1 = prob.factorize()
    n1, m1 = prob.matRows, prob.matCols
    n, m2 = J2.shape
    n2 = n - n1
    qp, qi, rp, ri = prob.pattern()
    Q1 = sp.csr_matrix((res1.Q_vals, qi, qp), shape=(n1, n1))
    R1 = sp.csc_matrix((res1.R_vals, ri, rp), shape=(n1, m1))
    J2t = np.array(J2, dtype=np.float64, order="F")
    J2t[:n1, :] = Q1.T @ J2t[:n1, :]
    bottom = np.asfortranarray(J2t[m1:, :])
    if right_solver == COLPIV:
        qr2, hc2, p2, _ = colpiv_qr(bottom)
    else:
        qr2, hc2 = householder_qr(bottom)
        p2 = np.arange(m2, dtype=np.int32)
    k2 = min(bottom.shape)
    R2 = np.triu(qr2[:k2, :])
    out = BAResult()
    out.left, out.Q1, out.R1 = res1, Q1, R1
    out.J2 = J2t
    out.qr2, out.hc2, out.P2 = qr2, hc2, p2
    top = sp.hstack([R1[:m1, :], sp.csc_matrix(J2t[:m1, :][:, p2])], format="csc")
    mid = sp.hstack([sp.csc_matrix((k2, m1)), sp.csc_matrix(R2)], format="csc")
    pad = sp.csc_matrix((n - m1 - k2, m1 + m2))
    out.R = sp.vstack([top, mid, pad], format="csc")
    out.perm = np.concatenate([res1.perm, m1 + p2]).astype(np.int32)
    out.rank = res1.rank + k2
    out.m1, out.m2, out.n1, out.n2 = m1, m2, n1, n2
    return out


def ba_apply_qt(res: BAResult, v: np.ndarray) -> np.ndarray:
    """BlockAngularSparseQR_QProduct::evalTo, transpose branch (BlockAngularSparseQR.h:607-625)."""
    out = np.array(v, dtype=np.float64).reshape(res.n1 + res.n2, -1).copy()
    out[:res.n1, :] = res.Q1.T @ out[:res.n1, :]
    bot = out[res.m1:, :]
    # Q2^T = H_{k-1} ... H_0 applied in order
    for k in range(len(res.hc2)):
        vk = np.concatenate([[1.0], res.qr2[k + 1:, k]])
        bot[k:, :] -= res.hc2[k] * np.outer(vk, vk @ bot[k:, :])
    out[res.m1:, :] = bot
    return out if np.ndim(v) > 1 else out[:, 0]


# ---------------------------------------------------------------- BandedBlockedSparseQR

class BBResult:
    pass


def bb_analyze(J, suggested: int = 2):
    """BandedBlockedSparseQR::analyzePattern, generic path (src/QRKit/BandedBlockedSparseQR.h:409-427):
    AsBandedAsPossible row ordering, then BlockBandedMatrixInfo::operator() on the permuted row-major matrix.
    Returns (row_perm, blocks); (P*M).row(row_perm[i]) = M.row(i)."""
    import scipy.sparse as sp
    M = sp.csr_matrix(J)
    M.sort_indices()
    has, perm = as_banded_as_possible(M.shape[0], M.shape[1], M.indptr, M.indices)
    if has:
        inv = np.empty_like(perm); inv[perm] = np.arange(len(perm), dtype=perm.dtype)
        M = M[inv]
        M.sort_indices()
    else:
        perm = np.arange(M.shape[0], dtype=np.int32)
    blocks = block_info_from_csr(M.shape[0], M.shape[1], M.indptr, M.indices, suggested)
    return perm, blocks


def bb_factorize(J, suggested: int = 2) -> BBResult:
    """BandedBlockedSparseQR::factorize (BandedBlockedSparseQR.h:443-519) restated with numpy on top of the C
    kernels (orc_householder_qr, orc_block_triangular_factor): sequential chain of dense panels, each the
    leftover triangle of the previous panel stacked on the next block rows; R rows emitted per panel incl. explicit
    zeros (:484-491); Y (unit lower) and negated T per panel (:471-481) with (row, numZeros) for the implicit Q."""
    import scipy.sparse as sp
    perm, blocks = bb_analyze(J, suggested)
    assert blocks is not None, "reference would hit undefined behaviour (mergeBlocks on an empty vector)"
    inv = np.empty_like(perm); inv[perm] = np.arange(len(perm), dtype=perm.dtype)
    pmat = sp.csr_matrix(J)[inv].toarray()           # m_pmat = m_rowPerm * mat  (:446)
    rows, cols = pmat.shape
    nb = len(blocks)
    Rd = np.zeros((rows, cols))
    Rmask = np.zeros((rows, cols), dtype=bool)
    yty = []
    bi = blocks[0]
    Ji = pmat[bi[0]:bi[0] + bi[2], bi[1]:bi[1] + bi[3]].copy()
    activeRows, numZeros = int(bi[2]), 0
    for i in range(nb):
        bi = blocks[i]
        idxRow, idxCol, numRows, numCols = (int(v) for v in bi)
        qr, hc = householder_qr(Ji)                                           # houseqr.compute(Ji) (:468)
        Y = np.eye(activeRows, numCols)
        for bc in range(numCols):
            Y[bc + 1:, bc] = qr[bc + 1:, bc]                                  # essentialVector(bc) (:472-475)
        T = -block_triangular_factor(Y, hc[:numCols])                         # (:476-477)
        yty.append((Y, T, idxCol, numZeros))                                  # BlockYTY(Y, T, diagIdx, diagIdx, numZeros) (:480-481)
        V = np.triu(qr)
        solved = numRows if i == nb - 1 else int(blocks[i + 1][1]) - idxCol   # (:486)
        for br in range(solved):
            Rd[idxCol + br, idxCol:idxCol + numCols] = V[br, :numCols] if br < V.shape[0] else 0.0
            Rmask[idxCol + br, idxCol:idxCol + numCols] = True
        if i < nb - 1:
            nx = blocks[i + 1]
            nRow, nCol, nRows, nCols = (int(v) for v in nx)
            overlap = (idxCol + numCols) - nCol
            colInc = numCols - overlap
            activeRows = numRows + nRows - colInc
            numZeros = max((nRow + nRows) - activeRows - nCol, 0)
            ncols = nCols if nCols >= overlap else overlap
            Ji = pmat[idxRow + colInc:idxRow + colInc + activeRows, nCol:nCol + ncols].copy()
            if overlap > 0:
                lr = activeRows - nRows
                Ji[:lr, :overlap] = V[colInc:colInc + lr, colInc:colInc + overlap]
    out = BBResult()
    out.row_perm, out.blocks, out.yty = perm, blocks, yty
    rr, cc = np.nonzero(Rmask)
    order = np.lexsort((rr, cc))
    out.R = sp.csc_matrix((Rd[rr[order], cc[order]], (rr[order], cc[order])), shape=(rows, cols))
    out.R_dense = Rd
    out.R_mask = Rmask
    out.rank = cols
    return out


def bb_apply_q(res: BBResult, v: np.ndarray, transpose: bool) -> np.ndarray:
    """SparseBlockYTY_VecProduct::evalTo (src/QRKit/SparseBlockYTY.h:100-139) with the two-segment gather/scatter
    of BlockYTY_VecProduct (BlockYTY.h:152-172, SparseQRUtils.h:47-89)."""
    out = np.array(v, dtype=np.float64).reshape(len(res.row_perm), -1).copy()
    seq = res.yty if transpose else res.yty[::-1]
    for (Y, T, row, nz) in seq:
        m, n = Y.shape
        idx = np.concatenate([np.arange(row, row + n), np.arange(row + n + nz, row + n + nz + (m - n))])
        seg = out[idx, :]
        TT = T.T if transpose else T
        seg = seg + Y @ (TT @ (Y.T @ seg))
        out[idx, :] = seg
    return out if np.ndim(v) > 1 else out[:, 0]


# ---------------------------------------------------------------- BlockedThinDenseQR / BlockedThinSparseQR

class BTResult:
    pass


def column_density(J):
    """SparseQROrdering::ColumnDensity (src/QRKit/SparseQROrdering.h:21-50): columns stable-sorted by their number of
    nonzeros (ascending); returns the INDICES of the Eigen PermutationMatrix it builds, colpermIndices(origIdx) = sorted rank
    (:43-46).  As everywhere in Eigen, (A * P)(:, j) = A(:, indices[j])."""
    import scipy.sparse as sp
    M = sp.csc_matrix(J)
    nnz = np.diff(M.indptr)
    order = np.argsort(nnz, kind="stable")           # order[rank] = original column
    idx = np.empty(M.shape[1], dtype=np.int32)
    idx[order] = np.arange(M.shape[1], dtype=np.int32)
    return idx


def _panel_yt(qr, hc, ncols):
    """BlockedThinQRBase::computeBlockedRepresentation (BlockedThinQRBase.h:322-333): Y = unit-lower essential vectors,
    T = -make_block_householder_triangular_factor(Y, hCoeffs)."""
    nrows = qr.shape[0]
    Y = np.zeros((nrows, ncols))
    for bc in range(ncols):
        if bc < nrows:
            Y[bc, bc] = 1.0
            Y[bc + 1:, bc] = qr[bc + 1:, bc]
    T = -block_triangular_factor(Y, np.concatenate([hc, np.zeros(max(0, ncols - len(hc)))])[:ncols])
    return Y, T


def _update_mat(mat, r0, Y, T, c_from, c_to):
    """BlockedThinQRBase::updateMat (BlockedThinQRBase.h:309-319): mat(rows of the block, j) += Y (T^T (Y^T mat(..., j)))."""
    rows = slice(r0, r0 + Y.shape[0])
    blk = mat[rows, c_from:c_to]
    mat[rows, c_from:c_to] = blk + Y @ (T.T @ (Y.T @ blk))


def bt_dense_qr(A, block_cols: int = 2) -> BTResult:
    """BlockedThinDenseQR::compute (src/QRKit/BlockedThinDenseQR.h:104-176): panels of `block_cols` columns at
    (solvedCols, solvedCols), HouseholderQR of the panel, Y/T, block-reflector update of columns solvedCols.. of m_R in place;
    identity permutations (:139-142)."""
    R = np.array(A, dtype=np.float64, order="F")
    rows, cols = R.shape
    res = BTResult()
    res.blocks = []
    solved = 0
    while solved < cols:
        new = block_cols
        nrows = rows - solved
        if solved + new >= cols:                      # updateBlockInfo (:145-156)
            new = cols - solved
        qr, hc = householder_qr(R[solved:solved + nrows, solved:solved + new])
        Y, T = _panel_yt(qr, hc, new)
        res.blocks.append((solved, Y, T))
        _update_mat(R, solved, Y, T, solved, cols)
        solved += new
    res.R = R
    res.perm = np.arange(cols, dtype=np.int32)
    res.rowperm = np.arange(rows, dtype=np.int32)
    res.rank = cols
    res.rows, res.cols = rows, cols
    return res


def bt_sparse_qr(J, block_cols: int = 2) -> BTResult:
    """BlockedThinSparseQR::compute (src/QRKit/BlockedThinSparseQR.h:105-283): ColumnDensity column ordering and
    AsBandedAsPossible row ordering (:168-201), densify (:120), then per panel: rows from the sparsity of the panel's columns
    (updateBlockInfo, :203-238), ColPivHouseholderQR of the panel (a copy), nonzero / zero pivot column bookkeeping (:250-256),
    Y/T, update of columns idxCol.. of the dense matrix, R built column-wise from the rows above the diagonal position and the
    panel's packed QR (:271-279); colsPermutation = ColumnDensity permutation * Householder column permutation (:151-159)."""
    import scipy.sparse as sp
    M = sp.csc_matrix(J)
    rows, cols = M.shape
    cperm = column_density(M)
    pm = sp.csc_matrix(M[:, cperm])                   # m_pmat = mat * m_outputPerm_c
    R0 = sp.csr_matrix(pm); R0.sort_indices()
    has, rperm = as_banded_as_possible(rows, cols, R0.indptr, R0.indices)
    if not has:
        rperm = np.arange(rows, dtype=np.int32)
    inv = np.empty_like(rperm); inv[rperm] = np.arange(rows, dtype=rperm.dtype)
    pm = sp.csc_matrix(sp.csr_matrix(pm)[inv])        # m_pmat = m_rowPerm * m_pmat: row i moves to row rperm[i]
    pm.sort_indices()
    D = pm.toarray(order="F")                         # m_pmatDense
    Rout = np.zeros((rows, cols))
    res = BTResult()
    res.blocks = []
    nnz_idx, zero_idx = [], []
    nzp = 0                                           # m_nonzeroPivots
    solved, new_piv, prev_rows = 0, 0, 0
    while solved < cols:
        new = block_cols
        if solved + new >= cols:
            new = cols - solved
            nrows = rows - nzp
        else:
            biggest = 0
            for c in range(new):
                col = pm.indices[pm.indptr[solved + c]:pm.indptr[solved + c + 1]]
                end = int(col[-1]) if len(col) else 0
                biggest = max(biggest, end)
            nrows = biggest - nzp + 1
            if nrows < prev_rows - new_piv:
                nrows = prev_rows - new_piv
        r0, c0 = nzp, solved
        Ji = D[r0:r0 + nrows, c0:c0 + new].copy()
        qr, hc, p, nz = colpiv_qr(Ji)
        nnz_idx += [c0 + int(p[c]) for c in range(nz)]
        zero_idx += [c0 + int(p[c]) for c in range(nz, new)]
        Y, T = _panel_yt(qr, hc, new)
        res.blocks.append((r0, Y, T))
        _update_mat(D, r0, Y, T, c0, cols)
        for bc in range(new):
            Rout[:nzp, nzp + bc] = D[:nzp, c0 + int(p[bc])]
            for br in range(bc + 1):
                if br < qr.shape[0]:
                    Rout[nzp + br, nzp + bc] = qr[br, bc]
        new_piv = nz
        nzp += nz
        prev_rows = nrows
        solved += new
    house = np.array(nnz_idx + zero_idx, dtype=np.int32)
    res.R = Rout
    res.perm = cperm[house].astype(np.int32)          # (P1 * P2).indices[j] = P1.indices[P2.indices[j]]
    res.rowperm = rperm.astype(np.int32)
    res.rank = nzp
    res.rows, res.cols = rows, cols
    return res


def bt_apply_q(res: BTResult, v: np.ndarray, transpose: bool) -> np.ndarray:
    """SparseBlockYTY sequenceYTY (SparseBlockYTY.h:111-138) for the thin solvers: Q^T v applies the blocks in order with T^T,
    Q v in reverse order with T (numZeros = 0: one row segment per block)."""
    out = np.array(v, dtype=np.float64).reshape(res.rows, -1).copy()
    seq = res.blocks if transpose else res.blocks[::-1]
    for r0, Y, T in seq:
        seg = out[r0:r0 + Y.shape[0], :]
        out[r0:r0 + Y.shape[0], :] = seg + Y @ ((T.T if transpose else T) @ (Y.T @ seg))
    return out if np.ndim(v) > 1 else out[:, 0]
