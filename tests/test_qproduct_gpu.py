"""GPU: matrixQ() products of every solver: Q*v, Q^T*v with dense vectors / matrices, sparse
right-hand sides and Q materialised as a sparse matrix, against the explicit Q (block-diagonal solver) or the oracle's
product (compositions), plus the algebraic identities the reference's tests check (test-qrkit.cpp:219-252)."""
import numpy as np
import pytest
import scipy.sparse as sp

from helpers import rel_fro, seeded_tiles
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qa():
    import qrkit_amd
    return qrkit_amd


@pytest.fixture(scope="module")
def ctx(qa):
    return qa.Context(0)


@pytest.mark.parametrize("B,r,c,extra,qf", [(300, 7, 2, 0, 0), (300, 7, 2, 3, 0), (50, 32, 32, 0, 0), (40, 20, 9, 5, 1),
                                            (7, 100, 37, 0, 0),
                                            # the grouped kernels of round 5 (2 .. 64 lanes per tile): every filling of the last wavefront,
                                            # both Q formats, tiles of 33 .. 64 rows (one tile per wavefront, staged through LDS)
                                            (1, 7, 2, 0, 0), (33, 8, 6, 0, 0), (129, 8, 6, 2, 1), (17, 16, 16, 0, 0), (9, 13, 5, 0, 1),
                                            (3, 32, 32, 1, 0), (21, 40, 40, 0, 0), (10, 64, 48, 0, 0), (6, 64, 64, 4, 1), (130, 5, 1, 0, 0)])
def test_block_diagonal_q_times_b_on_device(qa, ctx, B, r, c, extra, qf):
    """qrk_bd_apply_q: matrixQ() * b with the explicit Q (FullQ [U|N] split and BlockDiagonalQ), trailing identity rows."""
    tiles = seeded_tiles(B + r, -1.0, 1.0, B * r * c)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    mat = qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles, rows=B * r + extra)
    qr = qa.BlockDiagonalSparseQR(qFormat=qf, context=ctx)
    qr.compute(mat)
    Q = qr.matrixQ()
    b = np.random.default_rng(1).uniform(-1, 1, (B * r + extra, 3))
    assert rel_fro(qr.applyQ(b), Q @ b) <= 1e-14
    assert rel_fro(qr.applyQ(b[:, 0]), Q @ b[:, 0]) <= 1e-14
    assert rel_fro(qr.applyQ(qr.applyQt(b)), b) <= 1e-13          # Q Q^T = I


def mixed_batch(qa, ctx):
    rng = np.random.default_rng(3)
    B = 60
    cols = rng.integers(1, 50, B).astype(np.int32)
    rows = (cols + rng.integers(0, 9, B)).astype(np.int32)
    tiles = seeded_tiles(5, -1.0, 1.0, int((rows.astype(np.int64) * cols).sum()))
    mat = qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles)
    qr = qa.BlockDiagonalSparseQR(context=ctx)
    qr.compute(mat)
    return qr


def test_block_diagonal_q_times_b_mixed_sizes(qa, ctx):
    qr = mixed_batch(qa, ctx)
    b = np.random.default_rng(2).uniform(-1, 1, qr.rows())
    assert rel_fro(qr.applyQ(b), qr.matrixQ() @ b) <= 1e-14


def banded_problem(num_vars=64, seed=1):
    rng = np.random.default_rng(seed)
    num_params = 2 * num_vars
    ii, jj = [], []
    for i in range(num_params):
        for j in range(2 * i, min(2 * i + 2, num_params)):
            for r in range(7):
                ii.append(7 * i + r); jj.append(j)
            if j < num_params - 2:
                ii.append(7 * i + 6); jj.append(j + 2)
    m = sp.csr_matrix((rng.uniform(0.5, 5.0, len(ii)), (ii, jj)), shape=(7 * num_vars, num_params))
    m.sort_indices()
    return m


def test_banded_matrix_q_expression(qa, ctx):
    J = banded_problem()
    slv = qa.BandedBlockedSparseQR(8, context=ctx)
    slv.compute(J)
    Q = slv.matrixQ()
    n = J.shape[0]
    Qd = Q @ np.eye(n)                                  # slvr.matrixQ() * I   (test-qrkit.cpp:221-222)
    Qtd = Q.transpose() @ np.eye(n)                     # .transpose() * I     (:224-225)
    assert rel_fro(Qtd, Qd.T) <= 1e-13
    assert np.abs(Qd.T @ Qd - np.eye(n)).max() <= 1e-13
    P = slv.rowsPermutation()
    PJ = np.zeros(J.shape); PJ[P, :] = J.toarray()
    R = slv.matrixR().toarray()
    assert rel_fro(Qd @ R, PJ) <= 1e-12                 # (:249)
    # sparse right-hand side -> sparse result; Q materialised as a sparse matrix
    S = sp.random(n, 5, density=0.05, random_state=3, format="csc")
    QS = Q @ S
    assert sp.issparse(QS) and rel_fro(QS.toarray(), Qd @ S.toarray()) <= 1e-13
    Qs = Q.toSparse()
    assert sp.issparse(Qs) and rel_fro(Qs.toarray(), Qd) <= 1e-14
    assert Q.adjoint().rows() == n and rel_fro((Q.T @ S).toarray(), Qd.T @ S.toarray()) <= 1e-13


def test_angular_and_thin_matrix_q_expression(qa, ctx):
    num_vars, m2 = 128, 40
    vals = orc.gen_uniform(1, 0.5, 5.0, num_vars * 14 + 7 * num_vars * m2)
    tiles = vals[:num_vars * 14]
    J2 = vals[num_vars * 14:].reshape(7 * num_vars, m2)
    prob = orc.BDProblem.uniform(num_vars, 7, 2, tiles)
    ref = orc.ba_factorize(prob, J2)
    left = qa.SparseBlockDiagonal.fromTiles(prob.rows, prob.cols, tiles)
    ba = qa.BlockAngularSparseQR(context=ctx)
    ba.compute(qa.BlockMatrix1x2(left, J2))
    Q = ba.matrixQ()
    b = np.random.default_rng(4).uniform(-1, 1, (7 * num_vars, 2))
    y = Q.transpose() @ b
    assert rel_fro(y, orc.ba_apply_qt(ref, b)) <= 1e-12
    assert rel_fro(Q @ y, b) <= 1e-13                   # Q (Q^T b) = b  (BlockAngularSparseQR.h:627-645)
    # thin solver
    A = np.random.default_rng(6).uniform(-1, 1, (500, 30))
    thin = qa.BlockedThinDenseQR(ctx, 2)
    thin.compute(A.copy())
    Qt = thin.matrixQ()
    R = np.zeros((500, 30)); R[:30, :] = thin.matrixR().cpu().numpy()
    assert rel_fro(Qt @ R, A) <= 1e-13
    assert rel_fro(Qt.transpose() @ A, R) <= 1e-13
