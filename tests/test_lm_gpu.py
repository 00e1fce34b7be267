"""GPU: the Levenberg-Marquardt use of the block-diagonal solver.

* LM-damped blocks: rowpermADiagLambda (test/test-utils.cpp:145-180) interleaves sqrt(lambda) I with the rows of J so that
  [J; sqrt(lambda) I] stays block diagonal with 9x2 blocks (known answer test-utils.cpp:254-274); the blocks are cut on
  the device, factorised, and checked against the oracle and against the damped normal equations.
* The LM caller contract (examples/ellipse_fitting.cpp:126-142, bench_sparse_qr_extra.cpp:249-357): a solver object whose
  analyzePattern() is done once and whose factorize() runs every iteration; the step uses matrixQ().adjoint()*f,
  matrixR() and colsPermutation() -- here through solve().
"""
import numpy as np
import pytest
import scipy.sparse as sp

from helpers import RTOL, oracle_factorize, rel_fro, seeded_tiles
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def rowperm_a_diag_lambda(J, lam):
    """rowpermADiagLambda (test-utils.cpp:145-180): the lambda row of column c goes right below the last nonzero of c."""
    J = J.tocsc()
    n_res, n_par = J.shape
    perm = np.zeros(n_res + n_par, dtype=np.int64)
    curr = 0
    for c in range(n_par):
        col_rows = J.indices[J.indptr[c]:J.indptr[c + 1]]
        last = col_rows[-1] if len(col_rows) else 0
        while curr <= last + c:
            perm[curr - c] = curr; curr += 1
        perm[n_res + c] = curr; curr += 1
    stacked = sp.vstack([J, np.sqrt(lam) * sp.identity(n_par)]).tocsr()
    inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
    out = stacked[inv]          # (P*M).row(perm[i]) = M.row(i)
    out.sort_indices()
    return out, perm


def test_lm_damped_blocks_9x2():
    import qrkit_amd
    B, lam = 256, 1e-3
    vals = seeded_tiles(1, 0.5, 5.0, B * 14)
    J = sp.block_diag([vals[i * 14:(i + 1) * 14].reshape(2, 7).T for i in range(B)], format="csc")
    A, perm = rowperm_a_diag_lambda(J, lam)
    # structure: the reference's known answer (9i, 2i, 9, 2), from the oracle's generic analysis
    blocks = orc.block_info_from_csr(A.shape[0], A.shape[1], A.indptr, A.indices, 2)
    i = np.arange(B)
    np.testing.assert_array_equal(blocks, np.stack([9 * i, 2 * i, np.full(B, 9), np.full(B, 2)], 1))
    ctx = qrkit_amd.Context(0)
    blk = qrkit_amd.SparseBlockDiagonal().fromBlockDiagonalPattern(A, 9, 2, context=ctx)      # cut on the device (CSR)
    qr = qrkit_amd.BlockDiagonalSparseQR(context=ctx)
    qr.compute(blk)
    rows, cols = np.full(B, 9, np.int32), np.full(B, 2, np.int32)
    tiles = np.concatenate([A[9 * k:9 * k + 9, 2 * k:2 * k + 2].toarray().ravel(order="F") for k in range(B)])
    prob, ref = oracle_factorize(rows, cols, tiles)
    np.testing.assert_array_equal(qr.colsPermutation(), ref.perm)
    assert rel_fro(qr.rValues().cpu().numpy(), ref.R_vals) <= RTOL
    assert rel_fro(qr.qValues().cpu().numpy(), ref.Q_vals) <= RTOL
    # the LM step: min |J dx - f|^2 + lambda |dx|^2  <=>  (J^T J + lambda I) dx = J^T f
    f = np.random.default_rng(3).uniform(-1, 1, J.shape[0])
    rhs = np.zeros(A.shape[0]); rhs[perm[:J.shape[0]]] = f          # [f; 0] through the same row permutation
    dx = qr.solve(rhs)
    assert rel_fro(dx, prob.solve(ref, rhs)) <= 1e-11
    N = (J.T @ J + lam * sp.identity(J.shape[1])).tocsc()
    assert np.linalg.norm(N @ dx - J.T @ f) <= 1e-10 * np.linalg.norm(J.T @ f)


def test_lm_loop_with_cached_pattern():
    """Independent curve fits y = exp(a t) + b (7 samples, 2 parameters each): the Jacobian is block diagonal with 7x2
    blocks, damped to 9x2.  analyzePattern() once, factorize() per iteration."""
    import qrkit_amd
    B = 500
    rng = np.random.default_rng(5)
    t = np.linspace(0.0, 1.0, 7)
    a_true, b_true = rng.uniform(-1, 1, B), rng.uniform(-1, 1, B)
    y = np.exp(a_true[:, None] * t) + b_true[:, None]
    ctx = qrkit_amd.Context(0)
    qr = qrkit_amd.BlockDiagonalSparseQR(context=ctx)
    rows, cols = np.full(B, 9, np.int32), np.full(B, 2, np.int32)
    a, b = np.zeros(B), np.zeros(B)
    lam = 1e-2

    def residual(a, b):
        return np.exp(a[:, None] * t) + b[:, None] - y          # (B, 7)

    cost = 0.5 * np.sum(residual(a, b) ** 2)
    analysed = False
    for it in range(30):
        r = residual(a, b)
        tiles = np.zeros((B, 2, 9))                               # column-major 9x2 tiles
        tiles[:, 0, :7] = t * np.exp(a[:, None] * t)              # d/da
        tiles[:, 1, :7] = 1.0                                     # d/db
        tiles[:, 0, 7] = np.sqrt(lam)                             # [J_i; sqrt(lambda) I_2]
        tiles[:, 1, 8] = np.sqrt(lam)
        mat = qrkit_amd.SparseBlockDiagonal.fromTiles(rows, cols, tiles.ravel())
        if not analysed:
            qr.analyzePattern(mat)
            plan_before = qr._plan.value
            analysed = True
        qr.factorize(mat)
        assert qr._plan.value == plan_before                      # the pattern analysis is reused
        rhs = np.zeros((B, 9)); rhs[:, :7] = -r
        dx = qr.solve(rhs.ravel()).reshape(B, 2)
        new_cost = 0.5 * np.sum(residual(a + dx[:, 0], b + dx[:, 1]) ** 2)
        if new_cost < cost:
            a, b, cost, lam = a + dx[:, 0], b + dx[:, 1], new_cost, lam * 0.1
        else:
            lam *= 10.0
        if cost < 1e-24:
            break
    assert np.max(np.abs(a - a_true)) < 1e-8 and np.max(np.abs(b - b_true)) < 1e-8
