"""GPU parity of SparseBlockDiagonal::fromBlockDiagonalPattern cut on the device (qrk_bd_tiles_from_sparse,
SURVEY 8 row a2) against the reference's definition: tile i = dense copy of mat.block(idxRow, idxCol, numRows,
numCols) for the block map of BlockBandedMatrixInfo::fromBlockDiagonalPattern (SparseBlockDiagonal.h:71-89,
SparseQRUtils.h:255-272; block map from the oracle's restatement).  Bit-exact: the values are copied."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from helpers import oracle_factorize, rel_fro, RTOL, seeded_tiles
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qa():
    import qrkit_amd
    return qrkit_amd


@pytest.fixture(scope="module")
def ctx(qa):
    return qa.Context(0)


def block_diag_matrix(B, r, c, seed, extra_rows=0, off_block=0, density=1.0):
    """Block-diagonal sparse matrix as the reference's tests build it (test-qrkit.cpp:60-75: a triplet per tile
    entry), optionally thinned out, with trailing rows and with entries outside the blocks."""
    vals = seeded_tiles(seed, 0.5, 5.0, B * r * c)
    rng = np.random.default_rng(seed)
    ii = np.repeat(np.arange(B), r * c) * r + np.tile(np.tile(np.arange(r), c), B)
    jj = np.repeat(np.arange(B), r * c) * c + np.tile(np.repeat(np.arange(c), r), B)
    keep = rng.random(vals.size) < density
    ii, jj, vv = ii[keep], jj[keep], vals[keep]
    if off_block:
        oi = rng.integers(0, B * r + extra_rows, off_block)
        oj = rng.integers(0, B * c, off_block)
        outside = (oi // r != oj // c) | (oi >= B * r)
        ii = np.concatenate([ii, oi[outside]]); jj = np.concatenate([jj, oj[outside]])
        vv = np.concatenate([vv, rng.uniform(-1, 1, int(outside.sum()))])
    return sp.coo_matrix((vv, (ii, jj)), shape=(B * r + extra_rows, B * c))


def expected_tiles(mat, r, c):
    blocks = orc.from_block_diagonal_pattern(mat.shape[0], mat.shape[1], r, c)   # (idxRow, idxCol, numRows, numCols)
    d = mat.tocsc()
    out = []
    for (r0, c0, nr, nc) in np.asarray(blocks).reshape(-1, 4)[:, :4]:
        out.append(d[r0:r0 + nr, c0:c0 + nc].toarray().ravel(order="F"))
    return np.concatenate(out) if out else np.zeros(0)


@pytest.mark.parametrize("B,r,c,extra,off,density", [
    (1000, 32, 32, 0, 0, 1.0),      # BASELINE configs[0] shape
    (256, 7, 2, 0, 0, 1.0),         # the reference's test shape (test-qrkit.cpp:369-377)
    (300, 8, 6, 5, 200, 0.7),       # zeros inside the blocks, entries outside them, trailing rows
    (3000, 7, 2, 3, 500, 0.8),      # many small blocks: the thread-per-outer-index kernel (round 5), with zeros, strays, trailing rows
    (1500, 8, 6, 0, 0, 1.0), (1100, 16, 16, 2, 300, 0.6), (2000, 5, 1, 0, 100, 1.0),
    (40, 100, 37, 0, 50, 0.9),
    (6, 200, 150, 3, 0, 0.5),       # pieces: 200 x 20 columns at a time
    (2, 2000, 3, 0, 10, 0.8),       # the tallest tiles a plan takes: two columns per piece
    (17, 1, 1, 0, 0, 1.0),
])
@pytest.mark.parametrize("fmt", ["csc", "csr"])
def test_tiles_cut_on_device_are_the_dense_blocks(qa, ctx, B, r, c, extra, off, density, fmt):
    mat = block_diag_matrix(B, r, c, seed=3, extra_rows=extra, off_block=off, density=density).asformat(fmt)
    blk = qa.SparseBlockDiagonal().fromBlockDiagonalPattern(mat, r, c, context=ctx)
    assert blk.size() == B and blk.rows() == mat.shape[0] and blk.cols() == mat.shape[1]
    assert blk.tiles is None and blk.tiles_dev is not None and blk.tiles_dev.is_cuda   # cut on the device
    got = blk.tiles_dev.cpu().numpy()
    np.testing.assert_array_equal(got, expected_tiles(mat, r, c))


def test_host_space_entry_point_and_mixed_layout(qa, ctx):
    """QRK_MEM_HOST staging, and a mixed layout: base_row/base_col are the plan's running sums."""
    from qrkit_amd import _capi as capi
    rng = np.random.default_rng(11)
    B = 37
    cols = rng.integers(1, 70, B).astype(np.int32)
    rows = (cols + rng.integers(0, 40, B)).astype(np.int32)
    R, Cn = int(rows.sum()), int(cols.sum())
    dense = np.zeros((R + 4, Cn))
    r0 = c0 = 0
    want = []
    for nr, nc in zip(rows, cols):
        t = rng.uniform(-1, 1, (nr, nc)) * (rng.random((nr, nc)) < 0.8)
        dense[r0:r0 + nr, c0:c0 + nc] = t
        want.append(t.ravel(order="F"))
        r0 += nr; c0 += nc
    dense[R:, :] = 1.0                          # outside every block
    want = np.concatenate(want)
    for fmt, rm in (("csc", 0), ("csr", 1)):
        m = sp.coo_matrix(dense).asformat(fmt)
        m.sort_indices()
        lay = capi.BDLayout()
        lay.num_blocks, lay.mat_rows, lay.mat_cols = B, R + 4, Cn
        lay.rows = rows.ctypes.data_as(C.POINTER(C.c_int32))
        lay.cols = cols.ctypes.data_as(C.POINTER(C.c_int32))
        plan = C.c_void_p()
        capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), 0, 0, C.byref(plan)), ctx.handle)
        ptr = np.ascontiguousarray(m.indptr, np.int32); idx = np.ascontiguousarray(m.indices, np.int32)
        vals = np.ascontiguousarray(m.data, np.float64)
        out = np.full(want.size, np.nan)
        capi.check(capi.lib().qrk_bd_tiles_from_sparse(plan, rm, ptr.ctypes.data, idx.ctypes.data, vals.ctypes.data,
                                                       int(m.nnz), out.ctypes.data, capi.MEM_HOST), ctx.handle)
        capi.lib().qrk_bd_plan_destroy(plan)
        np.testing.assert_array_equal(out, want)


def test_reference_flow_sparse_to_solution(qa, ctx):
    """test-qrkit.cpp:167-206 end to end on the device: sparse J -> fromBlockDiagonalPattern -> compute -> invariants."""
    B, r, c = 256, 7, 2
    mat = block_diag_matrix(B, r, c, seed=1).tocsc()
    blk = qa.SparseBlockDiagonal().fromBlockDiagonalPattern(mat, r, c, context=ctx)
    qr = qa.BlockDiagonalSparseQR(context=ctx)
    qr.compute(blk)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    _, ref = oracle_factorize(rows, cols, expected_tiles(mat, r, c))
    np.testing.assert_array_equal(qr.colsPermutation(), ref.perm)
    assert rel_fro(qr.rValues().cpu().numpy(), ref.R_vals) <= RTOL
    assert rel_fro(qr.qValues().cpu().numpy(), ref.Q_vals) <= RTOL


def test_from_sparse_matrix_row_shuffled(qa, ctx):
    """SparseBlockDiagonal::fromSparseMatrix (SparseBlockDiagonal.h:95-130) on a row-shuffled block-diagonal matrix
    (the input of the reference's test 1, test-utils.cpp:182-209): the ordering and the block map come from the
    library's host analysis, the blocks are cut on the device.  SuggestedBlockCols = 3 merges pairs of 7x2 blocks."""
    B = 64
    mat = block_diag_matrix(B, 7, 2, seed=1).tocsr()
    p = np.random.default_rng(0).permutation(mat.shape[0])
    shuffled = mat[p]
    blk, row_perm = qa.SparseBlockDiagonal().fromSparseMatrix(shuffled, context=ctx)
    # oracle: the reference's ordering + block detection restated in C
    S = shuffled.copy(); S.sort_indices()
    has, operm = orc.as_banded_as_possible(S.shape[0], S.shape[1], S.indptr, S.indices)
    assert has
    np.testing.assert_array_equal(row_perm, operm)
    inv = np.empty_like(operm); inv[operm] = np.arange(len(operm))
    Sp = S[inv]; Sp.sort_indices()
    oblocks = orc.block_info_from_csr(Sp.shape[0], Sp.shape[1], Sp.indptr, Sp.indices, 3)
    np.testing.assert_array_equal(blk.block_rows, oblocks[:, 2])
    np.testing.assert_array_equal(blk.block_cols, oblocks[:, 3])
    want = np.concatenate([Sp[r0:r0 + nr, c0:c0 + nc].toarray().ravel(order="F") for r0, c0, nr, nc in oblocks])
    np.testing.assert_array_equal(blk.tiles_dev.cpu().numpy(), want)
    # the permuted matrix is the block-diagonal one again (rows within a block may be ordered differently)
    assert blk.rows() == mat.shape[0] and blk.cols() == mat.shape[1]
    qr = qa.BlockDiagonalSparseQR(context=ctx)
    qr.compute(blk)
    x = np.random.default_rng(1).uniform(-1, 1, mat.shape[1])
    b = np.empty(mat.shape[0]); b[row_perm] = shuffled @ x          # rowsPermutation * (J x)
    assert rel_fro(qr.solve(b), x) <= 1e-10


def test_from_sparse_matrix_rejects_banded(qa, ctx):
    B = 32
    m = block_diag_matrix(B, 7, 2, seed=2).tolil()
    for i in range(B - 1):
        m[7 * i + 6, 2 * i + 2] = 1.0          # overlap into the next block's columns
    with pytest.raises(ValueError):
        qa.SparseBlockDiagonal().fromSparseMatrix(m.tocsr(), context=ctx)
