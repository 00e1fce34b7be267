"""GPU: the dense right-block solver (qrk_dense_*: Eigen ColPivHouseholderQR / HouseholderQR with implicit Q, the
_BlockQRSolverRight of BlockAngularSparseQR, src/QRKit/BlockAngularSparseQR.h:361-369) against the oracle, on both
device paths: the single-workgroup kernel (dense_qr.hip), the row-slab path over all CUs (dense_qr_tall.hip) and the
column-parallel kernel (dense_qr_cols.hip: one launch per reflector, a wavefront per column)."""
import os

import numpy as np
import pytest

from helpers import rel_fro
from oracle import oracle as orc


def _factor(A, solver, path):
    import torch
    import qrkit_amd
    from qrkit_amd.angular import DenseColPivQR
    old = os.environ.get("QRK_DENSE_PATH")
    if path:
        os.environ["QRK_DENSE_PATH"] = path
    try:
        ctx = qrkit_amd.Context(0)
        qr = DenseColPivQR(ctx, solver)
        At = torch.from_numpy(np.asfortranarray(A).T.copy()).cuda().t()      # column-major storage on the device
        qr.compute(At)
        return qr, At
    finally:
        if old is None:
            os.environ.pop("QRK_DENSE_PATH", None)
        else:
            os.environ["QRK_DENSE_PATH"] = old


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,path", [(64, 20, "single"), (64, 20, "slabs"), (64, 20, "cols"), (300, 300, "slabs"),
                                             (300, 300, "cols"), (1000, 37, "single"), (1000, 37, "slabs"), (1000, 37, "cols"),
                                             (5000, 64, "slabs"), (5000, 64, "cols"), (2500, 130, "cols"), (30000, 24, None),
                                             (40, 60, "slabs"), (40, 60, "cols")])
@pytest.mark.parametrize("solver", [0, 1])
def test_dense_qr_matches_oracle(rows, cols, path, solver):
    rng = np.random.default_rng(rows * 7 + cols)
    A = rng.uniform(-1.0, 1.0, (rows, cols))
    if solver == 0 and cols >= 20:
        A[:, 3] = A[:, 1]                      # an exact tie of column norms and a rank deficiency
    qr, At = _factor(A, solver, path)
    got = At.cpu().numpy()
    k = min(rows, cols)
    if solver == 0:
        ref, hc, perm, _ = orc.colpiv_qr(A)
        np.testing.assert_array_equal(qr.colsPermutation().cpu().numpy(), perm)      # bit-exact
    else:
        ref, hc = orc.householder_qr(A)
        np.testing.assert_array_equal(qr.colsPermutation().cpu().numpy(), np.arange(cols))
    # (a duplicated column is a tie and then a rank collapse: the matrix is flagged and redone by the exact path, whose
    # result is the oracle's bit for bit -- noise below the last pivot included)
    if solver == 0 and cols >= 20:
        np.testing.assert_array_equal(got, ref)
    keep = k
    assert rel_fro(np.triu(got[:k, :])[:keep], np.triu(ref[:k, :])[:keep]) <= 1e-11
    assert rel_fro(qr._hc.cpu().numpy()[:keep], hc[:keep]) <= 1e-11
    assert rel_fro(np.tril(got, -1)[:, :keep], np.tril(ref, -1)[:, :keep]) <= 1e-10
    # Q^T A P = R and Q Q^T b = b through the implicit Q
    import torch
    P = qr.colsPermutation().cpu().numpy()
    B = torch.from_numpy(np.asfortranarray(A[:, P]).T.copy()).cuda().t()
    qr.applyQ(B, transpose=True)
    Rfull = np.zeros((rows, cols)); Rfull[:k, :] = np.triu(got[:k, :])
    assert rel_fro(B.cpu().numpy(), Rfull) <= 1e-11
    qr.applyQ(B, transpose=False)
    assert rel_fro(B.cpu().numpy(), A[:, P]) <= 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols", [(64, 20), (300, 300), (1000, 130)])
def test_dense_solve_r_on_device(rows, cols):
    """qrk_dense_solve_r: back substitution with the upper triangle of the packed QR (the R2 block of
    BlockAngularSparseQR::_solve_impl, :202-227) against a host triangular solve with the same R; several
    right-hand sides; and the whole least-squares solve x = P R^-1 (Q^T b)(0:cols)."""
    import scipy.linalg as sl
    import torch
    rng = np.random.default_rng(rows + cols)
    A = rng.uniform(-1.0, 1.0, (rows, cols))
    qr, _ = _factor(A, 0, None)
    R = qr.matrixR().cpu().numpy()[:cols, :cols]
    Y = rng.uniform(-1, 1, (cols, 3))
    want = sl.solve_triangular(R, Y, lower=False)
    B = torch.from_numpy(Y.T.copy()).cuda().t()                      # column-major on the device
    qr.solveR(B)
    torch.cuda.synchronize()
    assert rel_fro(B.cpu().numpy(), want) <= 1e-10 * max(1.0, np.linalg.cond(R) * 1e-6)
    x = rng.uniform(-1, 1, cols)
    b = torch.from_numpy((A @ x)[None, :].copy()).cuda().t()
    qr.applyQ(b, transpose=True)
    z = b[:cols, :].t().contiguous().t()
    qr.solveR(z)
    xs = np.empty(cols); xs[qr.colsPermutation().cpu().numpy()] = z.cpu().numpy()[:, 0]
    assert rel_fro(xs, x) <= 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("spins", [None, "0", "3"])
def test_dense_solve_r_many_workgroups_and_its_finishing_kernel(spins, monkeypatch):
    """n >= 512: the triangular solve runs a workgroup per block of 64 rows, each waiting on the flags of the blocks below.  A wait that
    runs out must never let a block go on with x that is not there: the block stops, the blocks above it stop, and the kernel queued
    behind solves them on one workgroup from the untouched rows of b.  QRK_SOLVE_R_SPINS=0 makes every block but the bottom one stop (the
    whole solve is then the finishing kernel's), 3 some of them; the answers are those of the one-workgroup kernel and of the host."""
    import scipy.linalg as sl
    import torch
    rows, cols = 900, 840
    rng = np.random.default_rng(9)
    A = rng.uniform(-1.0, 1.0, (rows, cols))
    qr, _ = _factor(A, 0, None)
    R = qr.matrixR().cpu().numpy()[:cols, :cols]
    Y = rng.uniform(-1, 1, (cols, 4))
    want = sl.solve_triangular(R, Y, lower=False)
    tol = 1e-10 * max(1.0, np.linalg.cond(R) * 1e-6)
    monkeypatch.setenv("QRK_SOLVE_R_COOP", "0")
    B0 = torch.from_numpy(Y.T.copy()).cuda().t()
    qr.solveR(B0); torch.cuda.synchronize()
    monkeypatch.delenv("QRK_SOLVE_R_COOP")
    if spins is not None:
        monkeypatch.setenv("QRK_SOLVE_R_SPINS", spins)
    B = torch.from_numpy(Y.T.copy()).cuda().t()
    qr.solveR(B); torch.cuda.synchronize()
    assert rel_fro(B.cpu().numpy(), want) <= tol
    assert rel_fro(B.cpu().numpy(), B0.cpu().numpy()) <= 1e-12 * max(1.0, np.linalg.cond(R) * 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,sparse_in", [(2000, 96, False), (7168, 384, False), (500, 40, True)])
def test_blocked_thin_dense_qr_matches_oracle(rows, cols, sparse_in):
    """BlockedThinDenseQR / BlockedThinSparseQR (BlockedThinDenseQR.h:104-176; 7168 x 384 is the right block of the
    reference's tests 5-6, test-qrkit.cpp:388-407): R, the reflectors, identity permutations, rank = cols, solve()."""
    import scipy.sparse as sp
    import torch
    import qrkit_amd
    rng = np.random.default_rng(rows + cols)
    A = rng.uniform(0.5, 5.0, (rows, cols))
    if sparse_in:
        A *= rng.random((rows, cols)) < 0.3
    ctx = qrkit_amd.Context(0)
    qr = qrkit_amd.BlockedThinDenseQR(ctx, 2)
    qr.compute(sp.csc_matrix(A) if sparse_in else A)
    ref, hc = orc.householder_qr(A)
    assert qr.rank() == cols and qr.rows() == rows and qr.cols() == cols
    np.testing.assert_array_equal(qr.colsPermutation().cpu().numpy(), np.arange(cols))
    np.testing.assert_array_equal(qr.rowsPermutation().cpu().numpy(), np.arange(rows))
    assert rel_fro(qr.matrixR().cpu().numpy(), np.triu(ref[:cols, :])) <= 1e-11
    assert rel_fro(qr._hc.cpu().numpy(), hc) <= 1e-11
    x = rng.uniform(-1, 1, cols)
    got = qr.solve(torch.from_numpy(A @ x).cuda()).cpu().numpy().ravel()
    assert rel_fro(got, x) <= 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("solver", [0, 1])
def test_dense_qr_persistent_form_matches_kernel_sequence(solver):
    """QRK_DENSE_PERSISTENT=1: the same steps inside one cooperative kernel with grid barriers (opt-in: measured slower).
    Bitwise the same result as the kernel sequence - same arithmetic, same order."""
    rng = np.random.default_rng(12)
    A = rng.uniform(-1.0, 1.0, (3000, 150))
    A[:, 7] = A[:, 2]
    out = []
    for flag in ("1", None):
        if flag:
            os.environ["QRK_DENSE_PERSISTENT"] = flag
        try:
            qr, At = _factor(A, solver, "slabs")
        finally:
            os.environ.pop("QRK_DENSE_PERSISTENT", None)
        out.append((At.cpu().numpy().copy(), qr._hc.cpu().numpy().copy(), qr.colsPermutation().cpu().numpy().copy()))
    np.testing.assert_array_equal(out[0][2], out[1][2])
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["pm1", "zero_one", "small_int", "circulant", "dup_cols"])
@pytest.mark.parametrize("rows,cols,path", [(2000, 40, "single"), (2000, 40, "slabs"), (300, 120, "slabs"), (2000, 40, "cols"),
                                             (300, 120, "cols")])
def test_dense_tie_battery_is_bitwise_the_oracle(kind, rows, cols, path):
    """A dense block whose pivot decisions are ties in exact arithmetic (or noise after a rank collapse): the fast kernels flag
    it and the dense exact path (bdqr_exact.hip, dense_exact_kernel) redoes it from the plan's copy of the input in Eigen's
    operation order -- permutation, packed QR (R and the essential vectors) and tau bit-identical to the oracle."""
    from test_ties_gpu import tie_tiles
    A = tie_tiles(kind, 1, rows, cols, seed=rows + cols).reshape(cols, rows).T.copy()
    qr, At = _factor(A, 0, path)
    ref, hc, perm, _ = orc.colpiv_qr(A)
    np.testing.assert_array_equal(qr.colsPermutation().cpu().numpy(), perm)
    got, k = At.cpu().numpy(), min(rows, cols)
    if not np.array_equal(got, ref):
        # not flagged: every decision was clear of rounding (e.g. indicator columns with distinct counts) -- the fast path's
        # result stands, within the tolerance
        assert rel_fro(np.triu(got[:k]), np.triu(ref[:k])) <= 1e-12
        assert rel_fro(np.tril(got, -1), np.tril(ref, -1)) <= 1e-11
        assert rel_fro(qr._hc.cpu().numpy()[:k], hc[:k]) <= 1e-12
    else:
        np.testing.assert_array_equal(qr._hc.cpu().numpy()[:k], hc[:k])
    if kind in ("pm1", "circulant", "dup_cols"):
        assert np.array_equal(got, ref), "a matrix with exact ties must have gone through the exact path"


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["slabs", "cols"])
def test_dense_generic_matrix_is_not_redone(path):
    """Generic data must stay on the fast path: its packed QR differs from the oracle's in the last bits (FMA chains)."""
    rng = np.random.default_rng(4)
    A = rng.uniform(-1.0, 1.0, (3000, 64))
    qr, At = _factor(A, 0, path)
    ref, hc, perm, _ = orc.colpiv_qr(A)
    np.testing.assert_array_equal(qr.colsPermutation().cpu().numpy(), perm)
    got = At.cpu().numpy()
    assert rel_fro(np.triu(got[:64]), np.triu(ref[:64])) <= 1e-12
    assert not np.array_equal(got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols", [(2048, 96), (4100, 200), (6000, 300), (700, 64)])
def test_two_stage_tall_colpiv_matches_oracle(rows, cols, monkeypatch):
    """The two-stage form of the pivoted QR of a tall right block (caqr.hip: A = Q0 R0 without pivoting on the matrix cores,
    then R0 P = Q1 R on the triangle) against Eigen's direct ColPivHouseholderQR as restated by the oracle
    (BlockAngularSparseQR.h:361-369): permutation bit-exact, R equal after aligning the sign of each row (the sign of a
    Householder beta follows the pivot entry, which the change of basis alters: SURVEY.md section 7), Q^T A P = R, Q Q^T b = b and
    the least-squares solution through the implicit Q.  Shapes: whole chunks / ragged last chunk and last panel / one slab only."""
    import torch
    monkeypatch.setenv("QRK_DENSE_TWO_STAGE", "1")
    rng = np.random.default_rng(rows + 3 * cols)
    A = rng.uniform(-1.0, 1.0, (rows, cols)) * rng.uniform(0.5, 2.0, cols)[None, :]
    qr, At = _factor(A, 0, None)
    got = At.cpu().numpy()
    ref, hc, perm, _ = orc.colpiv_qr(A)
    P = qr.colsPermutation().cpu().numpy()
    np.testing.assert_array_equal(P, perm)                                           # bit-exact
    # (the reflectors below the diagonal are those of Q0, not Eigen's: evidence that the two-stage path is the one that ran)
    assert rel_fro(np.tril(got, -1), np.tril(ref, -1)) > 1e-3
    Rg, Rr = np.triu(got[:cols, :]), np.triu(ref[:cols, :])
    sg = np.sign(np.diag(Rg)) * np.sign(np.diag(Rr))
    assert np.all(sg != 0)
    row_err = np.linalg.norm(Rg * sg[:, None] - Rr, axis=1) / np.linalg.norm(Rr, axis=1)
    assert row_err.max() <= 1e-11, row_err.max()                                     # every row of R, not one batch ratio
    B = torch.from_numpy(np.asfortranarray(A[:, P]).T.copy()).cuda().t()
    qr.applyQ(B, transpose=True)
    Rfull = np.zeros((rows, cols)); Rfull[:cols, :] = Rg
    assert np.linalg.norm(B.cpu().numpy() - Rfull) <= 1e-12 * np.linalg.norm(A) * np.sqrt(cols)
    qr.applyQ(B, transpose=False)
    assert rel_fro(B.cpu().numpy(), A[:, P]) <= 1e-12 * np.sqrt(cols)
    x = rng.uniform(-1, 1, cols)
    b = torch.from_numpy((A @ x)[None, :].copy()).cuda().t()
    qr.applyQ(b, transpose=True)
    assert np.linalg.norm(b.cpu().numpy()[cols:, 0]) <= 1e-11 * np.linalg.norm(A @ x)  # b is in the range of A
    z = b[:cols, :].t().contiguous().t()
    qr.solveR(z)
    xs = np.empty(cols); xs[P] = z.cpu().numpy()[:, 0]
    assert rel_fro(xs, x) <= 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["pm1", "dup_cols"])
def test_two_stage_tie_matrix_falls_back_to_the_exact_path(kind, monkeypatch):
    """A tall block with exact ties under the two-stage form: the pivoted second stage flags its decisions, the exact path redoes the
    WHOLE block from the plan's copy of the input in Eigen's operation order and leaves Eigen's packed format -- bitwise the oracle --
    and the products that follow must read that format (not the two-stage one)."""
    import torch
    from test_ties_gpu import tie_tiles
    monkeypatch.setenv("QRK_DENSE_TWO_STAGE", "1")
    rows, cols = 1536, 64
    A = tie_tiles(kind, 1, rows, cols, seed=7).reshape(cols, rows).T.copy()
    qr, At = _factor(A, 0, None)
    ref, hc, perm, _ = orc.colpiv_qr(A)
    np.testing.assert_array_equal(qr.colsPermutation().cpu().numpy(), perm)
    np.testing.assert_array_equal(At.cpu().numpy(), ref)
    np.testing.assert_array_equal(qr._hc.cpu().numpy()[:cols], hc[:cols])
    P = perm
    B = torch.from_numpy(np.asfortranarray(A[:, P]).T.copy()).cuda().t()
    qr.applyQ(B, transpose=True)
    Rfull = np.zeros((rows, cols)); Rfull[:cols, :] = np.triu(ref[:cols, :])
    assert np.linalg.norm(B.cpu().numpy() - Rfull) <= 1e-11 * np.linalg.norm(A)
    qr.applyQ(B, transpose=False)
    assert rel_fro(B.cpu().numpy(), A[:, P]) <= 1e-11


@pytest.mark.gpu
def test_two_stage_many_right_hand_sides_and_larger_block(monkeypatch):
    """Two-stage form on a larger block (12 288 x 448: 48 slabs, three levels, a ragged last panel of 0 columns -- 14 whole panels)
    with 70 right-hand sides (not a multiple of the 16-column MFMA tile): permutation and R against the oracle, Q^T B against
    the oracle's own Q^T B on the rows that are determined (the first `cols`, up to the row signs of R), Q (Q^T B) = B."""
    import torch
    monkeypatch.setenv("QRK_DENSE_TWO_STAGE", "1")
    rows, cols, nrhs = 12288, 448, 70
    rng = np.random.default_rng(99)
    A = rng.uniform(-1.0, 1.0, (rows, cols)) * rng.uniform(0.25, 4.0, cols)[None, :]
    qr, At = _factor(A, 0, None)
    ref, hc, perm, _ = orc.colpiv_qr(A)
    np.testing.assert_array_equal(qr.colsPermutation().cpu().numpy(), perm)
    Rg, Rr = np.triu(At.cpu().numpy()[:cols, :]), np.triu(ref[:cols, :])
    sg = np.sign(np.diag(Rg)) * np.sign(np.diag(Rr))
    row_err = np.linalg.norm(Rg * sg[:, None] - Rr, axis=1) / np.linalg.norm(Rr, axis=1)
    assert row_err.max() <= 1e-11, row_err.max()
    Bh = rng.uniform(-1.0, 1.0, (rows, nrhs))
    B = torch.from_numpy(np.asfortranarray(Bh).T.copy()).cuda().t()
    qr.applyQ(B, transpose=True)
    got = B.cpu().numpy()
    # the oracle's Q^T B through its own reflectors (rows 0..cols-1 are R^-T (A P)^T B: determined up to the sign of each row)
    want = orc.apply_householder_qt(ref, hc, Bh) if hasattr(orc, "apply_householder_qt") else None
    if want is None:
        AP = A[:, perm]
        want_top = np.linalg.solve(Rr.T, AP.T @ Bh)                     # = (Q^T B)(0:cols, :) for the oracle's R
        assert np.linalg.norm(got[:cols] * sg[:, None] - want_top) <= 1e-9 * np.linalg.norm(want_top)
    else:
        assert np.linalg.norm(got[:cols] * sg[:, None] - want[:cols]) <= 1e-10 * np.linalg.norm(want[:cols])
    assert abs(np.linalg.norm(got) - np.linalg.norm(Bh)) <= 1e-11 * np.linalg.norm(Bh)       # orthogonal
    qr.applyQ(B, transpose=False)
    assert rel_fro(B.cpu().numpy(), Bh) <= 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(10))
def test_random_dense_blocks_match_oracle(seed):
    """Randomised sweep over dense right blocks: tall, square and wide shapes (1..700 rows, 1..160 columns), both solvers, whichever
    device path the plan picks; generic values (fast path: <= 1e-11) or small integers (ties: exact path, bitwise)."""
    rng = np.random.default_rng(500 + seed)
    rows, cols = int(rng.integers(1, 701)), int(rng.integers(1, 161))
    solver = seed % 2
    ints = seed % 5 == 4
    A = rng.integers(-2, 3, (rows, cols)).astype(np.float64) if ints else rng.uniform(-1.0, 1.0, (rows, cols))
    qr, At = _factor(A, solver, None)
    got, k = At.cpu().numpy(), min(rows, cols)
    if solver == 0:
        ref, hc, perm, _ = orc.colpiv_qr(A)
    else:
        ref, hc = orc.householder_qr(A)
        perm = np.arange(cols)
    np.testing.assert_array_equal(qr.colsPermutation().cpu().numpy(), perm)
    if ints and solver == 0 and np.array_equal(got, ref):
        np.testing.assert_array_equal(qr._hc.cpu().numpy()[:k], hc[:k])
    else:
        scale = np.linalg.norm(np.triu(ref[:k]))
        assert np.linalg.norm(np.triu(got[:k]) - np.triu(ref[:k])) <= 1e-11 * max(scale, 1e-300)
        live = np.abs(np.diag(ref[:k, :k])) > 1e-9 * np.abs(ref[0, 0]) if k > 0 else np.zeros(0, bool)
        cols_live = np.where(live)[0]
        if cols_live.size:
            lo_g, lo_r = np.tril(got, -1)[:, cols_live], np.tril(ref, -1)[:, cols_live]
            assert np.linalg.norm(lo_g - lo_r) <= 1e-9 * max(np.linalg.norm(lo_r), 1.0)


@pytest.mark.gpu
def test_two_stage_structurally_orthogonal_columns_stay_on_the_fast_path(monkeypatch):
    """Column groups with disjoint row supports (cameras of a bundle adjustment that share no point): R0 of the first stage has
    entries at the noise level, so the pivoted second stage meets leading entries x0 ~ 1e-17 |A| whose sign - and with it the sign of
    beta and of a row of R - is rounding.  The two-stage R equals Eigen's only up to row signs anyway: no reason for the exact path
    (which took 200 s on the 40 000 x 2 000 block of BASELINE configs[3] with a camera-structured right block)."""
    import torch
    monkeypatch.setenv("QRK_DENSE_TWO_STAGE", "1")
    rows, cols, groups = 4096, 120, 12
    rng = np.random.default_rng(11)
    A = rng.uniform(0.5, 5.0, (rows, cols))
    A *= (np.arange(rows)[:, None] % groups) == (np.arange(cols)[None, :] // (cols // groups))
    qr, At = _factor(A, 0, None)
    got = At.cpu().numpy()
    ref, hc, perm, _ = orc.colpiv_qr(A)
    P = qr.colsPermutation().cpu().numpy()
    np.testing.assert_array_equal(P, perm)
    assert rel_fro(np.tril(got, -1), np.tril(ref, -1)) > 1e-3          # the two-stage format: the exact path did not run
    Rg, Rr = np.triu(got[:cols, :]), np.triu(ref[:cols, :])
    sg = np.sign(np.diag(Rg)) * np.sign(np.diag(Rr))
    assert np.all(sg != 0)
    row_err = np.linalg.norm(Rg * sg[:, None] - Rr, axis=1) / np.linalg.norm(Rr, axis=1)
    assert row_err.max() <= 1e-11, row_err.max()
    B = torch.from_numpy(np.asfortranarray(A[:, P]).T.copy()).cuda().t()
    qr.applyQ(B, transpose=True)
    Rfull = np.zeros((rows, cols)); Rfull[:cols, :] = Rg
    assert np.linalg.norm(B.cpu().numpy() - Rfull) <= 1e-12 * np.linalg.norm(A) * np.sqrt(cols)
    qr.applyQ(B, transpose=False)
    assert rel_fro(B.cpu().numpy(), A[:, P]) <= 1e-12 * np.sqrt(cols)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,rows,cols,two_stage,solver", [
    ("pm1", 4096, 128, "1", 0), ("dup_cols", 4096, 128, "0", 0), ("circulant", 3000, 100, "0", 0), ("pm1", 300, 70, "0", 0),
    ("pm1", 40, 60, "0", 0), ("small_int", 2111, 130, "0", 1)])
def test_dense_exact_path_over_the_whole_chip_is_bitwise_the_oracle(kind, rows, cols, two_stage, solver, monkeypatch):
    """Large dense blocks whose decisions are ties: the exact path as a host-launched sequence over all CUs (bdqr_exact.hip,
    launch_dense_exact_wide: a thread per column runs Eigen's sequential chains, one head and one apply launch per reflector) instead
    of one workgroup (200 s at 40 000 x 2 000): packed QR, tau and permutation bit-identical to the oracle, after the two-stage
    form and after the direct kernels, tall and landscape, pivoted and not (QRK_EXACT_WIDE=1 forces it on the small shapes)."""
    from test_ties_gpu import tie_tiles
    monkeypatch.setenv("QRK_DENSE_TWO_STAGE", two_stage)
    monkeypatch.setenv("QRK_EXACT_WIDE", "1")
    A = tie_tiles(kind, 1, rows, cols, seed=rows + cols).reshape(cols, rows).T.copy()
    qr, At = _factor(A, solver, None)
    k = min(rows, cols)
    if solver == 0:
        ref, hc, perm, _ = orc.colpiv_qr(A)
        np.testing.assert_array_equal(qr.colsPermutation().cpu().numpy(), perm)
    else:
        ref, hc = orc.householder_qr(A)[:2]
    got = At.cpu().numpy()
    if solver == 0:
        np.testing.assert_array_equal(got, ref)
        np.testing.assert_array_equal(qr._hc.cpu().numpy()[:k], hc[:k])
    else:
        # (un-pivoted: nothing but a degenerate reflector or a sign at the noise level flags the block; either it was redone - bitwise -
        #  or the fast result stands)
        assert np.array_equal(got, ref) or rel_fro(np.triu(got[:k]), np.triu(ref[:k])) <= 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols", [(6000, 300), (4100, 200), (16000, 520)])
def test_caqr_schedules_agree_bitwise(rows, cols, monkeypatch):
    """The first stage of the two-stage form runs on one, two (look-ahead) or three streams (look-ahead pipelined by levels,
    caqr.hip: caqr_factorize_pipelined).  The arithmetic does not depend on the schedule, so with the same kernels on every column
    (QRK_CAQR_NO_NARROW=1: the narrow apply kernel adds its partial sums in another order) the packed result, tau and the
    permutation must be bitwise the same in all three, run after run: a missing dependency between the streams shows up here."""
    import torch
    import qrkit_amd
    from qrkit_amd.angular import DenseColPivQR
    monkeypatch.setenv("QRK_DENSE_TWO_STAGE", "1")
    monkeypatch.setenv("QRK_CAQR_NO_NARROW", "1")
    ctx = qrkit_amd.Context(0)
    g = torch.Generator(device="cuda"); g.manual_seed(rows + cols)
    A0 = torch.rand((cols, rows), device="cuda", dtype=torch.float64, generator=g) * 4.5 + 0.5
    ref = None
    for mode in ("pipe", "plain", "none"):
        monkeypatch.setenv("QRK_CAQR_PIPE", "1" if mode == "pipe" else "0")
        monkeypatch.setenv("QRK_CAQR_LOOKAHEAD", "0" if mode == "none" else "1")
        qr = DenseColPivQR(ctx, 0)                      # (the switches are read when the plan is created)
        for _ in range(3 if mode == "pipe" else 1):
            At = A0.clone().t()
            qr.compute(At)
            torch.cuda.synchronize()
            res = (At.clone(), qr._hc.clone(), qr.colsPermutation().clone())
            if ref is None:
                ref = res
            for a, b in zip(res, ref):
                assert torch.equal(a, b), f"{mode}: the result depends on the schedule"


@pytest.mark.gpu
def test_two_stage_state_belongs_to_the_last_factorised_array(monkeypatch):
    """The two-stage format keeps part of Q (T of Q0, packed Q1) in the PLAN.  Factorising A and then B with one plan and applying
    Q_A afterwards must fail loudly instead of applying B's factors to A's reflectors; with the format switched off for the plan
    (qrk_dense_plan_set_two_stage(plan, 0): what BlockedThinSparseQR does for its cached per-shape plans) both Qs stay usable."""
    import ctypes as C
    import torch
    import qrkit_amd
    from qrkit_amd import _capi as capi
    monkeypatch.setenv("QRK_DENSE_TWO_STAGE", "1")
    rows, cols = 1024, 64
    rng = np.random.default_rng(5)
    ctx = qrkit_amd.Context(0)
    lib = capi.lib()
    plan = C.c_void_p()
    capi.check(lib.qrk_dense_plan_create(ctx.handle, rows, cols, capi.COLPIV_HOUSEHOLDER, C.byref(plan)), ctx.handle)
    assert lib.qrk_dense_plan_two_stage(plan) == 1
    mats, facs = [], []
    for _ in range(2):
        A = rng.uniform(-1.0, 1.0, (rows, cols))
        At = torch.from_numpy(np.asfortranarray(A).T.copy()).cuda().t()
        hc = torch.empty(cols, dtype=torch.float64, device="cuda")
        pp = torch.empty(cols, dtype=torch.int32, device="cuda")
        capi.check(lib.qrk_dense_factorize(plan, At.data_ptr(), rows, hc.data_ptr(), pp.data_ptr(), capi.MEM_DEVICE), ctx.handle)
        mats.append(A); facs.append((At, hc, pp))
    b = torch.from_numpy(rng.uniform(-1, 1, (1, rows))).cuda().t()
    # the plan now holds B's factors: A's are gone, and the call says so
    st = lib.qrk_dense_apply_q(plan, facs[0][0].data_ptr(), rows, facs[0][1].data_ptr(), 1, b.data_ptr(), rows, 1, capi.MEM_DEVICE)
    assert st == capi.STATUS_INVALID_ARGUMENT
    # ... while B's own product works
    P = facs[1][2].cpu().numpy()
    Bm = torch.from_numpy(np.asfortranarray(mats[1][:, P]).T.copy()).cuda().t()
    capi.check(lib.qrk_dense_apply_q(plan, facs[1][0].data_ptr(), rows, facs[1][1].data_ptr(), 1, Bm.data_ptr(), rows, cols, capi.MEM_DEVICE), ctx.handle)
    R = np.triu(facs[1][0].cpu().numpy()[:cols])
    assert np.linalg.norm(Bm.cpu().numpy()[:cols] - R) <= 1e-12 * np.linalg.norm(mats[1]) * np.sqrt(cols)
    # format off: self-contained factors, both usable afterwards
    capi.check(lib.qrk_dense_plan_set_two_stage(plan, 0), ctx.handle)
    facs = []
    for A in mats:
        At = torch.from_numpy(np.asfortranarray(A).T.copy()).cuda().t()
        hc = torch.empty(cols, dtype=torch.float64, device="cuda")
        pp = torch.empty(cols, dtype=torch.int32, device="cuda")
        capi.check(lib.qrk_dense_factorize(plan, At.data_ptr(), rows, hc.data_ptr(), pp.data_ptr(), capi.MEM_DEVICE), ctx.handle)
        facs.append((At, hc, pp))
    for A, (At, hc, pp) in zip(mats, facs):
        P = pp.cpu().numpy()
        Bm = torch.from_numpy(np.asfortranarray(A[:, P]).T.copy()).cuda().t()
        capi.check(lib.qrk_dense_apply_q(plan, At.data_ptr(), rows, hc.data_ptr(), 1, Bm.data_ptr(), rows, cols, capi.MEM_DEVICE), ctx.handle)
        ref, _, perm, _ = orc.colpiv_qr(A)
        np.testing.assert_array_equal(P, perm)
        assert rel_fro(np.triu(At.cpu().numpy()[:cols]), np.triu(ref[:cols])) <= 1e-12      # Eigen's format: signs included
        assert np.linalg.norm(Bm.cpu().numpy()[:cols] - np.triu(At.cpu().numpy()[:cols])) <= 1e-12 * np.linalg.norm(A) * np.sqrt(cols)
    lib.qrk_dense_plan_destroy(plan)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,force", [(2048, 96, True), (4100, 200, True), (20000, 256, False)])
def test_unpivoted_tall_block_runs_as_caqr(rows, cols, force, monkeypatch):
    """HouseholderQR of a tall dense block (the BlockedThinDenseQR chain, BlockedThinDenseQR.h:104-176, whose panel update is the
    block reflector of BlockedThinQRBase::updateMat, BlockedThinQRBase.h:309-333) as communication-avoiding QR on the matrix cores:
    R is the reference's up to the sign of each row (the elimination order differs), Q stays implicit.  The
    last shape takes the path by itself (rows >= 4 cols, cols >= 128, rows cols >= 2^22)."""
    import torch
    if force:
        monkeypatch.setenv("QRK_DENSE_TWO_STAGE", "1")
    rng = np.random.default_rng(rows + cols)
    A = rng.uniform(-1.0, 1.0, (rows, cols))
    qr, At = _factor(A, 1, None)
    got = At.cpu().numpy()
    ref, hc = orc.householder_qr(A)
    np.testing.assert_array_equal(qr.colsPermutation().cpu().numpy(), np.arange(cols))
    # (the reflectors below the diagonal are the tree's, not Eigen's: evidence that the CAQR path is the one that ran)
    assert rel_fro(np.tril(got, -1), np.tril(ref, -1)) > 1e-3
    Rg, Rr = np.triu(got[:cols, :]), np.triu(ref[:cols, :])
    sg = np.sign(np.diag(Rg)) * np.sign(np.diag(Rr))
    assert np.all(sg != 0)
    row_err = np.linalg.norm(Rg * sg[:, None] - Rr, axis=1) / np.linalg.norm(Rr, axis=1)
    assert row_err.max() <= 1e-11, row_err.max()
    B = torch.from_numpy(np.asfortranarray(A).T.copy()).cuda().t()
    qr.applyQ(B, transpose=True)
    Rfull = np.zeros((rows, cols)); Rfull[:cols, :] = Rg
    assert np.linalg.norm(B.cpu().numpy() - Rfull) <= 1e-12 * np.linalg.norm(A) * np.sqrt(cols)
    qr.applyQ(B, transpose=False)
    assert rel_fro(B.cpu().numpy(), A) <= 1e-12 * np.sqrt(cols)
    x = rng.uniform(-1, 1, cols)
    b = torch.from_numpy((A @ x)[None, :].copy()).cuda().t()
    qr.applyQ(b, transpose=True)
    z = b[:cols, :].t().contiguous().t()
    qr.solveR(z)
    assert rel_fro(z.cpu().numpy()[:, 0], x) <= 1e-9
