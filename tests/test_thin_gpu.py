"""The "thin" right-block solvers: BlockedThinDenseQR (src/QRKit/BlockedThinDenseQR.h:104-176) and
BlockedThinSparseQR (src/QRKit/BlockedThinSparseQR.h:105-283) -- the oracle's restatement of the reference's panel chains
against their own invariants (CPU), and the device solvers against the oracle (GPU)."""
import numpy as np
import pytest
import scipy.sparse as sp

from helpers import rel_fro
from oracle import oracle as orc


def thin_sparse_problem(rows, cols, seed, density=0.25):
    """A thin sparse matrix with a staircase profile (what the ordering is for) plus random fill, rows shuffled."""
    rng = np.random.default_rng(seed)
    M = sp.random(rows, cols, density=density, random_state=seed, format="lil", data_rvs=lambda n: rng.uniform(0.5, 5.0, n))
    for j in range(cols):
        M[min(rows - 1, j * (rows // cols)), j] = rng.uniform(0.5, 5.0)
    M = sp.csc_matrix(M)
    return sp.csc_matrix(M[rng.permutation(rows)])


def permuted(M, res_perm, res_rowperm):
    rows = M.shape[0]
    inv = np.empty(rows, dtype=np.int64); inv[np.asarray(res_rowperm)] = np.arange(rows)
    return sp.csr_matrix(M)[inv][:, np.asarray(res_perm)].toarray()


@pytest.mark.parametrize("rows,cols,bc", [(60, 13, 2), (200, 40, 3), (64, 64, 2)])
def test_oracle_thin_dense_chain(rows, cols, bc):
    A = np.random.default_rng(rows + cols).uniform(-1, 1, (rows, cols))
    r = orc.bt_dense_qr(A, bc)
    assert np.abs(np.tril(r.R, -1)).max() <= 1e-13
    assert rel_fro(orc.bt_apply_q(r, A, True), r.R) <= 1e-13                        # Q^T A = R
    assert rel_fro(orc.bt_apply_q(r, r.R, False), A) <= 1e-13                       # Q R = A
    qr, hc = orc.householder_qr(A)                                                  # the chain's reflectors are HouseholderQR's
    assert rel_fro(np.triu(r.R[:cols]), np.triu(qr[:cols])) <= 1e-13


@pytest.mark.parametrize("rows,cols,bc,seed", [(120, 20, 2, 1), (300, 48, 2, 2), (90, 30, 4, 3)])
def test_oracle_thin_sparse_chain(rows, cols, bc, seed):
    M = thin_sparse_problem(rows, cols, seed)
    s = orc.bt_sparse_qr(M, bc)
    assert sorted(s.perm.tolist()) == list(range(cols)) and sorted(s.rowperm.tolist()) == list(range(rows))
    PM = permuted(M, s.perm, s.rowperm)
    assert rel_fro(orc.bt_apply_q(s, PM, True), s.R) <= 1e-12                      # Q^T (Pr M Pc) = R
    assert np.abs(np.tril(s.R, -1)).max() == 0.0 and s.rank == cols
    x = np.random.default_rng(0).uniform(-1, 1, cols)
    y = orc.bt_apply_q(s, PM @ x, True)
    assert rel_fro(np.linalg.solve(s.R[:cols, :cols], y[:cols]), x) <= 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,bc", [(300, 40, 2), (5120, 384, 2), (64, 64, 3)])
def test_hip_thin_dense_matches_oracle_chain(rows, cols, bc):
    """The device factorisation (one HouseholderQR of the whole matrix, Q as essential vectors) against the oracle's
    restatement of the reference's panel chain: same R, same Q^T b."""
    import qrkit_amd
    A = np.random.default_rng(rows * 3 + cols).uniform(-1, 1, (rows, cols))
    ref = orc.bt_dense_qr(A, bc)
    ctx = qrkit_amd.Context(0)
    qr = qrkit_amd.BlockedThinDenseQR(ctx, bc)
    qr.compute(A)
    assert qr.rank() == ref.rank
    np.testing.assert_array_equal(qr.colsPermutation().cpu().numpy(), ref.perm)
    assert rel_fro(qr.matrixR().cpu().numpy(), np.triu(ref.R[:cols])) <= 1e-12
    b = np.random.default_rng(1).uniform(-1, 1, rows)
    assert rel_fro(qr._applyAny(b, True), orc.bt_apply_q(ref, b, True)) <= 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,bc,seed", [(120, 20, 2, 1), (300, 48, 2, 2), (90, 30, 4, 3), (2000, 96, 2, 4)])
def test_hip_thin_sparse_matches_oracle(rows, cols, bc, seed):
    """BlockedThinSparseQR on the device against the oracle: both permutations bit-exact, R and Q^T b within the tolerance,
    and the reference's use of it -- least-squares recovery through the permutations."""
    import qrkit_amd
    M = thin_sparse_problem(rows, cols, seed)
    ref = orc.bt_sparse_qr(M, bc)
    ctx = qrkit_amd.Context(0)
    qr = qrkit_amd.BlockedThinSparseQR(ctx, bc)
    qr.compute(M)
    assert qr.rank() == ref.rank
    np.testing.assert_array_equal(qr.colsPermutation().cpu().numpy(), ref.perm)        # bit-exact
    np.testing.assert_array_equal(qr.rowsPermutation().cpu().numpy(), ref.rowperm)     # bit-exact
    assert rel_fro(qr.matrixR().cpu().numpy(), ref.R) <= 1e-12
    b = np.random.default_rng(1).uniform(-1, 1, rows)
    assert rel_fro(qr._applyAny(b, True), orc.bt_apply_q(ref, b, True)) <= 1e-12
    assert rel_fro(qr._applyAny(qr._applyAny(b, True), False), b) <= 1e-13
    x = np.random.default_rng(0).uniform(-1, 1, cols)
    PM = permuted(M, ref.perm, ref.rowperm)
    assert rel_fro(qr.solve(PM @ x), x) <= 1e-9


@pytest.mark.gpu
def test_hip_thin_sparse_same_shape_panels_under_forced_two_stage(monkeypatch):
    """All panels of the chain share one dense plan and are re-applied later (matrixQ, solve).  With the two-stage format forced on
    (QRK_DENSE_TWO_STAGE=1) the plan would keep only the LAST panel's T / Q1: the chain switches the format off for its plan
    (qrk_dense_plan_set_two_stage), or the earlier panels would be applied with the wrong Q (round-2 advisor finding)."""
    import qrkit_amd
    monkeypatch.setenv("QRK_DENSE_TWO_STAGE", "1")
    rows, cols, bc = 96, 12, 4
    rng = np.random.default_rng(11)
    M = sp.csc_matrix(rng.uniform(0.5, 5.0, (rows, cols)))
    ref = orc.bt_sparse_qr(M, bc)
    ctx = qrkit_amd.Context(0)
    qr = qrkit_amd.BlockedThinSparseQR(ctx, bc)
    qr.compute(M)
    np.testing.assert_array_equal(qr.colsPermutation().cpu().numpy(), ref.perm)
    assert rel_fro(qr.matrixR().cpu().numpy(), ref.R) <= 1e-12
    b = rng.uniform(-1, 1, rows)
    assert rel_fro(qr._applyAny(b, True), orc.bt_apply_q(ref, b, True)) <= 1e-12
    assert rel_fro(qr._applyAny(qr._applyAny(b, True), False), b) <= 1e-13


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,bc,seed", [(150, 24, 2, 5), (200, 30, 3, 6)])
def test_hip_thin_sparse_rank_deficient_matches_oracle(rows, cols, bc, seed):
    """A rank-deficient sparse right block: an EMPTY column (an exact zero pivot: a column that is only numerically dependent leaves a
    pivot of rounding noise, whose size -- and with it Eigen's nonzero-pivot count -- is not reproducible between two correct
    implementations).  Rank, both permutations (zero-pivot columns last, BlockedThinSparseQR.h:151-159) and R against the oracle."""
    import qrkit_amd
    M = sp.lil_matrix(thin_sparse_problem(rows, cols, seed))
    M[:, 11] = 0.0
    M = sp.csc_matrix(M)
    M.eliminate_zeros()
    ref = orc.bt_sparse_qr(M, bc)
    assert ref.rank == cols - 1
    ctx = qrkit_amd.Context(0)
    qr = qrkit_amd.BlockedThinSparseQR(ctx, bc)
    qr.compute(M)
    assert qr.rank() == ref.rank
    np.testing.assert_array_equal(qr.colsPermutation().cpu().numpy(), ref.perm)
    np.testing.assert_array_equal(qr.rowsPermutation().cpu().numpy(), ref.rowperm)
    R = qr.matrixR().cpu().numpy()
    assert rel_fro(R, ref.R) <= 1e-12
    # the leading rank columns are a QR of the nonzero columns: Q^T (Pr M Pc)(:, 0:rank) = R(:, 0:rank)
    rk = ref.rank
    PM = permuted(M, ref.perm, ref.rowperm)
    QtPM = qr._applyAny(PM[:, :rk], True)
    assert np.linalg.norm(QtPM[:cols] - R[:cols, :rk]) <= 1e-12 * np.linalg.norm(PM) * np.sqrt(cols)
    assert np.linalg.norm(QtPM[cols:]) <= 1e-12 * np.linalg.norm(PM) * np.sqrt(cols)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,bc,seed", [(900, 300, 300, 7), (1200, 320, 400, 8), (700, 290, 272, 9)])
def test_hip_thin_sparse_wide_panels(rows, cols, bc, seed):
    """SuggestedBlockCols of more than 256 columns, and block_cols >= cols > 256 (ONE panel of the whole matrix): the diagonal and the
    permutation of a panel reach the host through a packing kernel that used to cover the first 256 entries only (round-3 advisor
    finding: garbage pivots and an out-of-bounds permutation beyond).  Banded-ish input as well: every panel runs at its own height
    (a bucket of it), not at the height of the whole matrix."""
    import qrkit_amd
    M = thin_sparse_problem(rows, cols, seed, density=0.05)
    ref = orc.bt_sparse_qr(M, bc)
    ctx = qrkit_amd.Context(0)
    qr = qrkit_amd.BlockedThinSparseQR(ctx, bc)
    qr.compute(M)
    assert qr.rank() == ref.rank
    np.testing.assert_array_equal(qr.colsPermutation().cpu().numpy(), ref.perm)
    np.testing.assert_array_equal(qr.rowsPermutation().cpu().numpy(), ref.rowperm)
    assert rel_fro(qr.matrixR().cpu().numpy(), ref.R) <= 1e-11
    b = np.random.default_rng(1).uniform(-1, 1, rows)
    assert rel_fro(qr._applyAny(b, True), orc.bt_apply_q(ref, b, True)) <= 1e-11
    assert rel_fro(qr._applyAny(qr._applyAny(b, True), False), b) <= 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_hip_thin_sparse_pivot_at_the_rank_threshold(seed):
    """nonzeroPivots() of a panel (BlockedThinSparseQR.h:250-256) is counted by Eigen from its DOWNDATED column norms against
    (largest initial norm * eps)^2 (rows - q) / rows -- not from |R_qq|, which equals the updated norm only up to ~1e-8 relative.  A panel
    whose last column is dependent up to a perturbation delta: delta is bisected ON THE ORACLE down to two adjacent doubles, one that counts
    3 nonzero pivots and one that counts 4; the device (whose exact path carries Eigen's own norm table) must agree on both sides, and a few
    ulp further out."""
    import qrkit_amd
    rng = np.random.default_rng(seed)
    rows, cols, bc = 24, 4, 4
    C = rng.uniform(0.5, 2.0, (rows, 3))
    w = rng.uniform(-1.0, 1.0, rows)

    def mat(delta):
        A = np.empty((rows, cols))
        A[:, :3] = C
        A[:, 3] = (C[:, 0] + C[:, 1]) + delta * w
        return sp.csc_matrix(A)

    lo, hi = 0.0, 1e-12                                     # rank 3 at delta = 0 (dependent up to rounding), 4 at 1e-12
    if orc.bt_sparse_qr(mat(lo), bc).rank != 3 or orc.bt_sparse_qr(mat(hi), bc).rank != 4:
        pytest.skip("the bracket does not hold for this seed")
    while np.nextafter(lo, np.inf) < hi:
        mid = 0.5 * (lo + hi)
        if mid == lo or mid == hi:
            break
        if orc.bt_sparse_qr(mat(mid), bc).rank == 3:
            lo = mid
        else:
            hi = mid
    ctx = qrkit_amd.Context(0)
    deltas = [lo, hi]
    for _ in range(4):
        deltas = [np.nextafter(deltas[0], 0.0)] + deltas + [np.nextafter(deltas[-1], np.inf)]
    seen = set()
    for d in deltas:
        M = mat(d)
        ref = orc.bt_sparse_qr(M, bc)
        qr = qrkit_amd.BlockedThinSparseQR(ctx, bc)
        qr.compute(M)
        assert qr.rank() == ref.rank, (d, qr.rank(), ref.rank)
        np.testing.assert_array_equal(qr.colsPermutation().cpu().numpy(), ref.perm)
        seen.add(ref.rank)
    assert seen == {3, 4}


def short_panel_problem(rows=120, cols=24, seed=11):
    """Columns 0..3 are the four sparsest and only have entries in the same two rows.  (ColumnDensity's permutation maps column j to the
    density RANK of column j, SparseQROrdering.h:43-46: the four columns of rank 0..3 stay in front exactly when they are columns 0..3.)
    With block_cols = 4 the first panel is 2 x 4 -- fewer rows than columns."""
    rng = np.random.default_rng(seed)
    M = sp.lil_matrix(thin_sparse_problem(rows, cols, seed, density=0.35))
    for j in (0, 1, 2, 3):
        M[:, j] = 0.0
        M[5, j] = rng.uniform(0.5, 5.0)
        M[77, j] = rng.uniform(0.5, 5.0)
    M = sp.csc_matrix(M)
    M.eliminate_zeros()
    return M


def test_oracle_thin_sparse_short_panel_is_in_the_fixture_shape():
    M = short_panel_problem()
    ref = orc.bt_sparse_qr(M, 4)
    assert ref.blocks[0][1].shape[0] == 2, "the first panel of the fixture is meant to have two rows"
    assert sorted(ref.perm.tolist()) == list(range(M.shape[1]))


@pytest.mark.gpu
def test_hip_thin_sparse_panel_with_fewer_rows_than_columns():
    """A panel with fewer rows than columns (round-4 advisor finding): Eigen's ColPivHouseholderQR stops after min(rows, cols) steps and
    leaves the other columns where its transpositions put them; the device factorises the panel zero-padded for all its columns, so the
    order of the tail is restored on the host from the first k choices.  Rank and both permutations against the oracle."""
    import qrkit_amd
    M = short_panel_problem()
    ref = orc.bt_sparse_qr(M, 4)
    qr = qrkit_amd.BlockedThinSparseQR(qrkit_amd.Context(0), 4)
    qr.compute(M)
    assert qr.rank() == ref.rank
    np.testing.assert_array_equal(qr.colsPermutation().cpu().numpy(), ref.perm)
    np.testing.assert_array_equal(qr.rowsPermutation().cpu().numpy(), ref.rowperm)
