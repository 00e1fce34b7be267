"""Block-angular composition (BlockAngularSparseQR with a block-diagonal left solver and a dense right solver):
the oracle against the reference's invariant (CPU) and the HIP path against the oracle (GPU)."""
import numpy as np
import pytest
import scipy.sparse as sp

from helpers import rel_fro
from oracle import oracle as orc


def angular_problem(num_vars, m2, seed=1, n2=0):
    """generate_block_angular_matrix shape (test/test-qrkit.cpp:135-165) with non-overlapping 7x2 blocks:
    J1 = num_vars blocks of 7x2, J2 = dense (7*num_vars + n2) x m2, all U(0.5, 5) from the reference's generator."""
    vals = orc.gen_uniform(seed, 0.5, 5.0, num_vars * 14 + (7 * num_vars + n2) * m2)
    tiles = vals[:num_vars * 14]
    J2 = vals[num_vars * 14:].reshape(7 * num_vars + n2, m2)      # row-major draw order: for i, for j (:155-159)
    prob = orc.BDProblem.uniform(num_vars, 7, 2, tiles)
    J1 = sp.block_diag([tiles[i * 14:(i + 1) * 14].reshape(2, 7).T for i in range(num_vars)], format="csc")
    if n2:
        J1 = sp.vstack([J1, sp.csc_matrix((n2, 2 * num_vars))], format="csc")
    return prob, tiles, J1, J2


def test_oracle_angular_invariants():
    """test_block_angular's assertion (LS recovery, test-qrkit.cpp:260-292) plus Q^T J P = R, on the oracle."""
    prob, tiles, J1, J2 = angular_problem(64, 24)
    res = orc.ba_factorize(prob, J2)
    J = sp.hstack([J1, sp.csc_matrix(J2)], format="csc")
    JP = J[:, res.perm].toarray()
    assert rel_fro(orc.ba_apply_qt(res, JP), res.R.toarray()) <= 1e-12
    x = np.random.default_rng(0).uniform(-1, 1, J.shape[1])
    y = orc.ba_apply_qt(res, J @ x)
    import scipy.linalg as sl
    solved = sl.solve_triangular(res.R[:J.shape[1], :].toarray(), y[:J.shape[1]])
    back = np.zeros_like(solved); back[res.perm] = solved
    assert rel_fro(back, x) <= 1e-10
    assert res.rank == J.shape[1]


@pytest.mark.gpu
@pytest.mark.parametrize("num_vars,m2,n2", [(64, 24, 0), (256, 96, 0), (1024, 384, 0), (100, 30, 17)])
def test_hip_angular_matches_oracle(num_vars, m2, n2):
    """Sizes up to the reference's own test (1024 blocks of 7x2 + 384 dense columns, test-qrkit.cpp:388-395)."""
    import qrkit_amd
    prob, tiles, J1, J2 = angular_problem(num_vars, m2, n2=n2)
    ref = orc.ba_factorize(prob, J2)
    left = qrkit_amd.SparseBlockDiagonal.fromTiles(prob.rows, prob.cols, tiles)
    ba = qrkit_amd.BlockAngularSparseQR()
    ba.compute(qrkit_amd.BlockMatrix1x2(left, J2))
    assert ba.info() == 0 and ba.rank() == ref.rank
    np.testing.assert_array_equal(ba.colsPermutation(), ref.perm)                   # bit-exact
    assert rel_fro(ba.matrixR().toarray(), ref.R.toarray()) <= 1e-12
    # LS recovery, the reference's own assertion (test-qrkit.cpp:289), through Q^T and through solve()
    J = sp.hstack([J1, sp.csc_matrix(J2)], format="csc")
    x = np.random.default_rng(0).uniform(-1, 1, J.shape[1])
    b = J @ x
    assert rel_fro(ba.applyQt(b), orc.ba_apply_qt(ref, b)) <= 1e-12
    assert rel_fro(ba.solve(b), x) <= 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("num_vars,m2", [(256, 96), (1024, 384)])
def test_hip_angular_with_thin_right_solver(num_vars, m2):
    """test_block_angular_denseblocked (test-qrkit.cpp:294-327): the right solver is BlockedThinDenseQR, i.e.
    Householder QR without column pivoting; P2 is the identity."""
    import qrkit_amd
    prob, tiles, J1, J2 = angular_problem(num_vars, m2)
    ref = orc.ba_factorize(prob, J2, right_solver=orc.NOPIV)
    left = qrkit_amd.SparseBlockDiagonal.fromTiles(prob.rows, prob.cols, tiles)
    ba = qrkit_amd.BlockAngularSparseQR(rightSolver=qrkit_amd.HOUSEHOLDER)
    ba.compute(qrkit_amd.BlockMatrix1x2(left, J2))
    assert ba.info() == 0 and ba.rank() == ref.rank
    np.testing.assert_array_equal(ba.colsPermutation(), ref.perm)                   # bit-exact
    m1 = J1.shape[1]
    np.testing.assert_array_equal(ba.colsPermutation()[m1:], np.arange(m1, m1 + m2))
    assert rel_fro(ba.matrixR().toarray(), ref.R.toarray()) <= 1e-12
    J = sp.hstack([J1, sp.csc_matrix(J2)], format="csc")
    x = np.random.default_rng(0).uniform(-1, 1, J.shape[1])
    assert rel_fro(ba.solve(J @ x), x) <= 1e-9


@pytest.mark.gpu
def test_configs3_composition_reduced_matches_oracle():
    """BASELINE configs[3] shape (tiles 8x6 + dense right block, concretised in SURVEY 8(d)) at 2000 tiles x 200 dense columns
    against the oracle's BlockAngularSparseQR::factorize (BlockAngularSparseQR.h:459-514): permutation bit-exact, R and Q^T b
    within the tolerance."""
    import qrkit_amd
    nt, m2 = 2000, 200
    vals = orc.gen_uniform(7, -1.0, 1.0, nt * 48 + 8 * nt * m2)
    tiles, J2 = vals[:nt * 48], vals[nt * 48:].reshape(8 * nt, m2)
    prob = orc.BDProblem.uniform(nt, 8, 6, tiles)
    ref = orc.ba_factorize(prob, J2)
    left = qrkit_amd.SparseBlockDiagonal.fromTiles(prob.rows, prob.cols, tiles)
    ba = qrkit_amd.BlockAngularSparseQR()
    ba.compute(qrkit_amd.BlockMatrix1x2(left, J2))
    assert ba.info() == 0 and ba.rank() == ref.rank
    np.testing.assert_array_equal(ba.colsPermutation(), ref.perm)                   # bit-exact
    R, Rref = ba.matrixR().tocsc(), ref.R.tocsc()
    assert abs(R - Rref).max() <= 1e-12 * abs(Rref).max()
    b = np.random.default_rng(1).uniform(-1, 1, 8 * nt)
    assert rel_fro(ba.applyQt(b), orc.ba_apply_qt(ref, b)) <= 1e-12


@pytest.mark.gpu
def test_configs3_composition_full_size_properties():
    """BASELINE configs[3] at FULL size as a composition: 20000 diagonal tiles of 8x6 (J1: 160000 x 120000) + a dense right
    block 160000 x 2000 (2.56 GB), one GPU.  Size-independent properties of BlockAngularSparseQR::factorize
    (BlockAngularSparseQR.h:459-514): valid permutation P = [P1; m1 + P2], rank, Q^T [J1 | J2] P = R on sampled columns of both
    parts, non-increasing |diag(R2)|, and the reference's own assertion -- least-squares recovery (test-qrkit.cpp:289)."""
    import torch
    import qrkit_amd
    nt, m2 = 20000, 2000
    n1, m1 = 8 * nt, 6 * nt
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    tiles = torch.rand(nt * 48, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    J2 = torch.rand(n1, m2, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    left = qrkit_amd.SparseBlockDiagonal.fromTiles(np.full(nt, 8, np.int32), np.full(nt, 6, np.int32), tiles)
    ba = qrkit_amd.BlockAngularSparseQR()
    ba.compute(qrkit_amd.BlockMatrix1x2(left, J2))
    assert ba.info() == 0 and ba.rank() == m1 + m2
    P = torch.as_tensor(ba.colsPermutation().astype(np.int64), device="cuda")
    assert bool((P.sort().values == torch.arange(m1 + m2, device="cuda")).all())
    assert bool((P[:m1] < m1).all()) and bool((P[m1:] >= m1).all())
    # sampled columns of J P through Q^T against the same columns of R
    A = tiles.view(nt, 6, 8)                                   # tile t, column c, row r (column-major tiles)
    cols = [0, 5, 77777, m1 - 1, m1, m1 + 1, m1 + 999, m1 + m2 - 1]
    JP = torch.zeros(n1, len(cols), device="cuda", dtype=torch.float64)
    for q, j in enumerate(cols):
        pj = int(P[j])
        if pj < m1:
            t, c = divmod(pj, 6)
            JP[8 * t:8 * t + 8, q] = A[t, c]
        else:
            JP[:, q] = J2[:, pj - m1]
    QtJP = ba.applyQt(JP)
    R = ba.matrixR().tocsc()
    Rs = torch.as_tensor(R[:, cols].toarray(), device="cuda")
    scale = float(Rs.abs().max())
    assert float((QtJP - Rs).abs().max()) <= 1e-11 * scale
    d2 = np.abs(R[m1:m1 + m2, m1:m1 + m2].diagonal())
    assert np.all(d2[1:] <= d2[:-1] * (1 + 1e-12))             # column pivoting of the right block
    # least-squares recovery
    x = torch.rand(m1 + m2, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    b = torch.bmm(A.transpose(1, 2), x[:m1].view(nt, 6, 1)).view(n1) + J2 @ x[m1:]
    xs = ba.solve(b)
    assert float((xs - x).norm() / x.norm()) <= 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_sparse_window_to_dense(fmt):
    """qrk_sparse_window_to_dense (the dense copy of a sparse right block written on the device, BlockedThinSparseQR.h:131): whole
    matrix, a row window, a window with a row map, an empty window, empty rows and columns - against scipy's toarray()."""
    import torch
    import qrkit_amd
    from qrkit_amd.angular import sparse_to_device_dense
    ctx = qrkit_amd.Context(0)
    rng = np.random.default_rng(5)
    for rows, cols, dens in ((300, 40, 0.2), (1000, 7, 0.5), (65, 130, 0.05), (5, 3, 1.0), (200, 16, 0.0)):
        M = sp.random(rows, cols, density=dens, random_state=int(rng.integers(1 << 30)), format=fmt, dtype=np.float64)
        D = M.toarray()
        out = sparse_to_device_dense(ctx, M)
        assert out.shape == (rows, cols) and out.t().is_contiguous()
        np.testing.assert_array_equal(out.cpu().numpy(), D)
        r0, nr = rows // 3, rows // 2
        np.testing.assert_array_equal(sparse_to_device_dense(ctx, M, r0, nr).cpu().numpy(), D[r0:r0 + nr])
        rmap = rng.permutation(nr).astype(np.int32)
        want = np.zeros((nr, cols)); want[rmap] = D[r0:r0 + nr]
        np.testing.assert_array_equal(sparse_to_device_dense(ctx, M, r0, nr, rmap).cpu().numpy(), want)
        assert sparse_to_device_dense(ctx, M, rows, 0).shape == (0, cols)
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_hip_angular_sparse_right_block(fmt):
    """A sparse right block (test-qrkit.cpp:335, BlockMatrix1x2<JacobianType, JacobianType>) gives bit for bit what its dense copy
    gives: the same dense J2 reaches the same kernels, only the way it reaches the device differs."""
    import qrkit_amd
    prob, tiles, J1, J2 = angular_problem(200, 48, n2=9)
    J2 = J2 * (np.random.default_rng(2).uniform(size=J2.shape) < 0.3)          # 70 % structural zeros
    left = qrkit_amd.SparseBlockDiagonal.fromTiles(prob.rows, prob.cols, tiles)
    dense, sparse = qrkit_amd.BlockAngularSparseQR(), qrkit_amd.BlockAngularSparseQR()
    dense.compute(qrkit_amd.BlockMatrix1x2(left, J2))
    sparse.compute(qrkit_amd.BlockMatrix1x2(left, sp.csr_matrix(J2) if fmt == "csr" else sp.csc_matrix(J2)))
    ref = orc.ba_factorize(prob, J2)
    assert sparse.rank() == dense.rank() == ref.rank
    np.testing.assert_array_equal(sparse.colsPermutation(), ref.perm)
    np.testing.assert_array_equal(sparse.matrixR().toarray(), dense.matrixR().toarray())
    assert rel_fro(sparse.matrixR().toarray(), ref.R.toarray()) <= 1e-12
    b = np.random.default_rng(1).uniform(-1, 1, J2.shape[0])
    np.testing.assert_array_equal(sparse.applyQt(b), dense.applyQt(b))
