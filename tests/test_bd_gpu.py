"""GPU parity tests of the block-diagonal hot path: HIP kernels (through the C ABI) vs the CPU oracle."""
import numpy as np
import pytest

from helpers import RTOL, oracle_factorize, per_tile_rel, rel_fro, seeded_tiles, tile_sizes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qa():
    import qrkit_amd
    return qrkit_amd


@pytest.fixture(scope="module")
def ctx(qa):
    return qa.Context(0)


def run_gpu(qa, ctx, rows, cols, tiles, mat_rows=None, q_format=0, solver=0, hc=True):
    mat = qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles, rows=mat_rows)
    qr = qa.BlockDiagonalSparseQR(blockSolver=solver, qFormat=q_format, context=ctx, hCoeffs=hc)
    qr.compute(mat)
    return mat, qr


def compare(qr, ref, rows, cols, tol=RTOL, hc=True):
    """Permutation bit-exact; Q, R, tau within `tol` PER TILE (largest per-tile relative Frobenius error)."""
    assert qr.info() == ref.info
    assert qr.rank() == ref.rank
    np.testing.assert_array_equal(qr.colsPermutation(), ref.perm)        # bit-exact
    sq, sr, sc = tile_sizes(rows, cols)
    nq = int(sq.sum())                                                    # (Q may carry trailing identity rows)
    assert per_tile_rel(qr.rValues().cpu().numpy(), ref.R_vals, sr) <= tol
    assert per_tile_rel(qr.qValues().cpu().numpy()[:nq], ref.Q_vals[:nq], sq) <= tol
    np.testing.assert_array_equal(qr.qValues().cpu().numpy()[nq:], ref.Q_vals[nq:])
    if hc:
        # tau of a degenerate last reflector is exactly 0 on both sides; elsewhere relative per tile
        assert per_tile_rel(qr.hCoeffs().cpu().numpy(), ref.hcoeffs, sc) <= tol


@pytest.mark.parametrize("B,r,c,lo,hi,seed", [
    (1000, 32, 32, 0.5, 5.0, 1),      # BASELINE configs[0]: 1000 blocks of 32x32
    (257, 32, 32, -1.0, 1.0, 2),
    (256, 7, 2, 0.5, 5.0, 1),         # the reference's own test shape (test-qrkit.cpp:369-377)
    (300, 8, 6, -1.0, 1.0, 3),
    (300, 6, 6, -1.0, 1.0, 3),
    (500, 2, 1, 0.5, 5.0, 1),
    (65, 32, 20, -1.0, 1.0, 4),
    (33, 20, 20, -1.0, 1.0, 5),
    (10, 1, 1, -1.0, 1.0, 6),
])
@pytest.mark.parametrize("solver", [0, 1])
@pytest.mark.parametrize("hc", [True, False])     # hc=False: the instantiation bench.py and the C++ facade run (tau not stored)
def test_uniform_tiles_match_oracle(qa, ctx, B, r, c, lo, hi, seed, solver, hc):
    tiles = seeded_tiles(seed, lo, hi, B * r * c)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    _, qr = run_gpu(qa, ctx, rows, cols, tiles, solver=solver, hc=hc)
    _, ref = oracle_factorize(rows, cols, tiles, block_solver=solver)
    compare(qr, ref, rows, cols, hc=hc)


def test_mixed_sizes_match_oracle(qa, ctx):
    rng = np.random.default_rng(7)
    B = 400
    cols = rng.integers(1, 33, B).astype(np.int32)
    rows = (cols + rng.integers(0, 6, B)).clip(max=32).astype(np.int32)
    n = int((rows.astype(np.int64) * cols).sum())
    tiles = seeded_tiles(11, -1.0, 1.0, n)
    for qf in (0, 1):
        for hc in (True, False):
            _, qr = run_gpu(qa, ctx, rows, cols, tiles, mat_rows=int(rows.sum()) + 5, q_format=qf, hc=hc)
            prob, ref = oracle_factorize(rows, cols, tiles, mat_rows=int(rows.sum()) + 5, q_format=qf)
            compare(qr, ref, rows, cols, hc=hc)
        for got, want in zip(qr.pattern(), prob.pattern()):
            np.testing.assert_array_equal(got, want)


def test_reference_invariants_7x2(qa, ctx):
    """The three invariants of test_block_diagonal (test/test-qrkit.cpp:201-203) on the reference's
    own input (generate_block_diagonal_matrix, 256 blocks of 7x2)."""
    from oracle import oracle as orc
    import scipy.sparse as sp
    nv = 256
    tiles = orc.gen_reference_7x2(nv)
    blocks = [tiles[i * 14:(i + 1) * 14].reshape(2, 7).T for i in range(nv)]
    J = sp.block_diag(blocks, format="csc")
    mat = qa.SparseBlockDiagonal().fromBlockDiagonalPattern(J, 7, 2)
    qr = qa.BlockDiagonalSparseQR(context=ctx)
    qr.compute(mat)
    Q, R, P = qr.matrixQ(), qr.matrixR(), qr.colsPermutation()
    JP = J[:, P]
    assert rel_fro((Q @ R).toarray(), JP.toarray()) <= 1e-13
    assert rel_fro((Q.T @ JP).toarray(), R.toarray()) <= 1e-13
    x = np.random.default_rng(0).uniform(-1, 1, J.shape[1])
    b = J @ x
    y = qr.applyQt(b)
    import scipy.linalg as sl
    solved = sp.linalg.spsolve_triangular(sp.csr_matrix(R[:J.shape[1], :]), y[:J.shape[1]], lower=False)
    back = np.zeros_like(solved)
    back[P] = solved
    assert rel_fro(back, x) <= 1e-11
    assert rel_fro(qr.solve(b), x) <= 1e-11


def test_solve_matches_oracle(qa, ctx):
    B, r, c = 200, 32, 32
    tiles = seeded_tiles(3, -1.0, 1.0, B * r * c)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    _, qr = run_gpu(qa, ctx, rows, cols, tiles)
    prob, ref = oracle_factorize(rows, cols, tiles)
    b = np.random.default_rng(1).uniform(-1, 1, (B * r, 3))
    x = qr.solve(b)
    xr = prob.solve(ref, b)
    assert rel_fro(x, xr) <= 1e-9    # conditioning of random 32x32 tiles amplifies 1e-16 rounding
    y = qr.applyQt(b[:, 0])
    Qd = qr.matrixQ()
    assert rel_fro(y, Qd.T @ b[:, 0]) <= 1e-13


def test_ties_and_rank_deficient_tiles(qa, ctx):
    """Exact ties must resolve to the first maximum (lowest current position), as Eigen's maxCoeff does."""
    r = c = 32
    rng = np.random.default_rng(5)
    t = []
    a = rng.uniform(-1, 1, (r, c)); a[:, 5] = a[:, 17]; a[:, 20] = a[:, 17]; t.append(a)     # duplicate columns
    a = rng.uniform(-1, 1, (r, c)); a[:, 3] = 0.0; a[:, 9] = 0.0; t.append(a)                 # zero columns
    t.append(np.eye(r))                                                                       # all norms equal
    t.append(np.zeros((r, c)))                                                                # all zero
    a = np.outer(rng.uniform(-1, 1, r), rng.uniform(-1, 1, c)); t.append(a)                   # rank one
    t.append(np.ones((r, c)))
    tiles = np.concatenate([x.ravel(order="F") for x in t])
    B = len(t)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    _, qr = run_gpu(qa, ctx, rows, cols, tiles)
    _, ref = oracle_factorize(rows, cols, tiles)
    P = qr.colsPermutation()
    # every one of these tiles meets a tie (or a zero norm) on its way: the fast kernel flags it and the exact path
    # (bdqr_exact.hip) redoes it in Eigen's operation order -- permutation, R and Q are the oracle's, bit for bit
    np.testing.assert_array_equal(P, ref.perm)
    np.testing.assert_array_equal(qr.rValues().cpu().numpy(), ref.R_vals)
    np.testing.assert_array_equal(qr.qValues().cpu().numpy(), ref.Q_vals)
    # every tile must still satisfy A P = Q R with orthogonal Q
    Qv = qr.qValues().cpu().numpy().reshape(B, r, r)
    Rv = qr.rValues().cpu().numpy().reshape(B, -1)
    for i in range(B):
        li = np.tril_indices(c); Rm = np.zeros((c, c)); Rm[li[1], li[0]] = Rv[i]   # packed upper triangle by columns
        Pi = P[i * c:(i + 1) * c] - i * c
        assert sorted(Pi.tolist()) == list(range(c))
        assert np.linalg.norm(Qv[i] @ np.vstack([Rm, np.zeros((r - c, c))]) - t[i][:, Pi]) <= 1e-13 * max(1.0, np.linalg.norm(t[i]))
        assert np.linalg.norm(Qv[i].T @ Qv[i] - np.eye(r)) <= 1e-13


def test_landscape_tile_is_invalid_input(qa, ctx):
    rows, cols = np.array([4, 3], np.int32), np.array([2, 5], np.int32)
    tiles = seeded_tiles(1, -1, 1, 4 * 2 + 3 * 5)
    mat = qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles)
    qr = qa.BlockDiagonalSparseQR(context=ctx)
    qr.compute(mat)
    assert qr.info() == qa.INFO_INVALID_INPUT
    assert not qr.m_isInitialized


def _dense_from_device(qr, B, r, c):
    import torch
    Q = qr.qValues().view(B, r, r)                       # row-major rows of Q_i
    Rv = qr.rValues().view(B, c * (c + 1) // 2)
    li = torch.tril_indices(c, c, device=Rv.device)
    R = torch.zeros(B, c, c, dtype=torch.float64, device=Rv.device)
    R[:, li[1], li[0]] = Rv                              # packed upper triangle by columns
    return Q, R


@pytest.mark.parametrize("B,r,c", [(10000, 32, 32), (20000, 8, 6), (20000, 6, 6)])
def test_full_size_properties(qa, ctx, B, r, c):
    """BASELINE configs[1] (10000 x 32x32) and the left stage of configs[3] at full size: size-independent
    properties checked on the device for every tile: A P = Q R, Q^T Q = I, valid permutations, plus a sampled
    comparison with the oracle."""
    import torch
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    tiles = torch.rand(B * r * c, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    mat = qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles)
    qr = qa.BlockDiagonalSparseQR(context=ctx)
    qr.compute(mat)
    assert qr.info() == 0 and qr.rank() == B * c
    Q, R = _dense_from_device(qr, B, r, c)
    A = tiles.view(B, c, r).transpose(1, 2)             # column-major tiles
    P = torch.as_tensor(qr.colsPermutation(), device="cuda").view(B, c).long() - (torch.arange(B, device="cuda") * c)[:, None]
    assert bool((P.sort(dim=1).values == torch.arange(c, device="cuda")[None, :]).all())
    AP = torch.gather(A, 2, P[:, None, :].expand(B, r, c))
    Rfull = torch.zeros(B, r, c, dtype=torch.float64, device="cuda"); Rfull[:, :c, :] = R
    err = (torch.bmm(Q, Rfull) - AP).flatten(1).norm(dim=1) / AP.flatten(1).norm(dim=1)
    assert float(err.max()) <= 1e-13
    orth = (torch.bmm(Q.transpose(1, 2), Q) - torch.eye(r, dtype=torch.float64, device="cuda")).flatten(1).norm(dim=1)
    assert float(orth.max()) <= 1e-13
    # |R_kk| non-increasing (column pivoting)
    d = R.diagonal(dim1=1, dim2=2).abs()
    assert bool((d[:, 1:] <= d[:, :-1] * (1 + 1e-12)).all())
    # sampled tiles against the oracle
    ns = 300
    _, ref = oracle_factorize(rows[:ns], cols[:ns], tiles[:ns * r * c].cpu().numpy())
    np.testing.assert_array_equal(qr.colsPermutation()[:ns * c], ref.perm)
    sq, sr, _ = tile_sizes(rows[:ns], cols[:ns])
    assert per_tile_rel(qr.qValues()[:ns * r * r].cpu().numpy(), ref.Q_vals, sq) <= RTOL
    assert per_tile_rel(qr.rValues()[:ns * (c * (c + 1) // 2)].cpu().numpy(), ref.R_vals, sr) <= RTOL


@pytest.mark.parametrize("B,r,c,seed", [(12, 64, 64, 1), (7, 100, 37, 2), (5, 33, 33, 3), (3, 256, 256, 4), (4, 200, 129, 5)])
@pytest.mark.parametrize("solver", [0, 1])
@pytest.mark.parametrize("hc", [True, False])
def test_large_tiles_match_oracle(qa, ctx, B, r, c, seed, solver, hc):
    """Tiles larger than 32x32 (workgroup-per-tile kernel), sizes of BASELINE configs[4] (8..256)."""
    tiles = seeded_tiles(seed, -1.0, 1.0, B * r * c)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    _, qr = run_gpu(qa, ctx, rows, cols, tiles, solver=solver, hc=hc)
    _, ref = oracle_factorize(rows, cols, tiles, block_solver=solver)
    compare(qr, ref, rows, cols, hc=hc)


def test_mixed_sizes_8_to_256_match_oracle(qa, ctx):
    """BASELINE configs[4] shape at reduced count: square tiles n ~ U{8..256}, mixed in one batch
    (small ones go to the half-wave kernel, large ones to the workgroup kernel)."""
    rng = np.random.default_rng(12345)
    B = 120
    n = rng.integers(8, 257, B).astype(np.int32)
    tiles = seeded_tiles(21, -1.0, 1.0, int((n.astype(np.int64) ** 2).sum()))
    _, qr = run_gpu(qa, ctx, n, n, tiles)
    prob, ref = oracle_factorize(n, n, tiles)
    compare(qr, ref, n, n)
    for got, want in zip(qr.pattern(), prob.pattern()):
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("B,r,c,nrhs", [(1, 7, 2, 1), (33, 7, 2, 2), (1000, 7, 2, 1), (257, 9, 2, 3), (65, 8, 6, 1), (400, 8, 6, 2), (31, 6, 6, 1),
                                        (70, 16, 16, 2), (19, 16, 9, 1), (5, 32, 32, 1), (101, 32, 20, 2), (40, 5, 1, 1), (12, 40, 24, 1)])
def test_solve_grouped_kernel_matches_oracle(qa, ctx, B, r, c, nrhs):
    """solve() of uniform batches of at most 32 columns (2 .. 32 lanes per tile, 64 / G tiles per wavefront, bd_aux.hip): every filling
    of the last wavefront, several right-hand sides, tall tiles; against the oracle's solve and per tile against numpy's least squares."""
    tiles = seeded_tiles(40 + B + r, 0.5, 5.0, B * r * c)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    _, qr = run_gpu(qa, ctx, rows, cols, tiles)
    prob, ref = oracle_factorize(rows, cols, tiles)
    b = np.random.default_rng(B).uniform(-1, 1, (B * r, nrhs))
    x = qr.solve(b if nrhs > 1 else b[:, 0])
    xr = prob.solve(ref, b if nrhs > 1 else b[:, 0])
    assert rel_fro(x, xr) <= 1e-9
    A = tiles.reshape(B, c, r).transpose(0, 2, 1)
    for i in (0, B // 2, B - 1):
        xi = np.linalg.lstsq(A[i], b[i * r:(i + 1) * r], rcond=None)[0]
        got = np.asarray(x).reshape(B * c, -1)[i * c:(i + 1) * c]
        assert np.linalg.norm(got - xi) <= 1e-9 * max(1.0, np.linalg.norm(xi))


def test_solve_ragged_small_tiles_matches_oracle(qa, ctx):
    rng = np.random.default_rng(77)
    rows = rng.integers(2, 33, 300).astype(np.int32)
    cols = np.minimum(rows, rng.integers(1, 33, 300)).astype(np.int32)
    tiles = seeded_tiles(78, -1.0, 1.0, int((rows.astype(np.int64) * cols).sum()))
    _, qr = run_gpu(qa, ctx, rows, cols, tiles)
    prob, ref = oracle_factorize(rows, cols, tiles)
    b = rng.uniform(-1, 1, (int(rows.sum()), 2))
    assert rel_fro(qr.solve(b), prob.solve(ref, b)) <= 1e-8


def test_solve_large_tiles(qa, ctx):
    B, r, c = 6, 150, 90
    tiles = seeded_tiles(8, -1.0, 1.0, B * r * c)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    _, qr = run_gpu(qa, ctx, rows, cols, tiles)
    prob, ref = oracle_factorize(rows, cols, tiles)
    b = np.random.default_rng(2).uniform(-1, 1, (B * r, 2))
    assert rel_fro(qr.solve(b), prob.solve(ref, b)) <= 1e-10


def test_mixed_8_to_256_properties_and_determinism(qa, ctx):
    """BASELINE configs[4] shape at 1/100 of its count (1000 square tiles, n ~ U{8..256}): the size-independent
    properties on every tile -- A P = Q R, Q^T Q = I, valid permutation, non-increasing |R_kk| -- and bitwise
    identical results when the same batch is factorised again (no atomics, no order dependence)."""
    rng = np.random.default_rng(2024)
    B = 1000
    n = rng.integers(8, 257, B).astype(np.int32)
    tiles = seeded_tiles(33, -1.0, 1.0, int((n.astype(np.int64) ** 2).sum()))
    _, qr = run_gpu(qa, ctx, n, n, tiles)
    assert qr.info() == 0 and qr.rank() == int(n.sum())
    Qv, Rv, perm = qr.qValues().cpu().numpy().copy(), qr.rValues().cpu().numpy().copy(), qr.colsPermutation().copy()
    toff = qoff = roff = coff = 0
    worst_qr = worst_orth = 0.0
    for i in range(B):
        c = int(n[i])
        A = tiles[toff:toff + c * c].reshape(c, c).T
        Q = Qv[qoff:qoff + c * c].reshape(c, c)
        li = np.tril_indices(c)
        R = np.zeros((c, c)); R[li[1], li[0]] = Rv[roff:roff + c * (c + 1) // 2]
        P = perm[coff:coff + c] - coff
        assert np.array_equal(np.sort(P), np.arange(c))
        worst_qr = max(worst_qr, np.linalg.norm(Q @ R - A[:, P]) / np.linalg.norm(A))
        worst_orth = max(worst_orth, np.linalg.norm(Q.T @ Q - np.eye(c)))
        d = np.abs(np.diag(R))
        assert np.all(d[1:] <= d[:-1] * (1 + 1e-12))
        toff += c * c; qoff += c * c; roff += c * (c + 1) // 2; coff += c
    assert worst_qr <= 1e-13 and worst_orth <= 1e-12, (worst_qr, worst_orth)
    _, qr2 = run_gpu(qa, ctx, n, n, tiles)
    np.testing.assert_array_equal(qr2.colsPermutation(), perm)
    np.testing.assert_array_equal(qr2.qValues().cpu().numpy(), Qv)
    np.testing.assert_array_equal(qr2.rValues().cpu().numpy(), Rv)


@pytest.mark.gpu
def test_edge_shapes_match_oracle():
    """Edge cases of the container and of the per-block solver: an empty matrix (no blocks), a single 1x1 tile, an all-zero tile
    (rank 0: every reflector degenerate, Q = I), a tile with one row more than columns, and a 1-column tile next to a wide-ish one."""
    import qrkit_amd
    ctx = qrkit_amd.Context(0)
    # no blocks at all: empty factors, empty permutation, Success
    empty = qrkit_amd.SparseBlockDiagonal.fromTiles(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0))
    qr = qrkit_amd.BlockDiagonalSparseQR(empty, context=ctx)
    assert qr.info() == 0 and qr.rank() == 0
    assert len(qr.colsPermutation()) == 0 and qr.rValues().numel() == 0 and qr.qValues().numel() == 0
    rng = np.random.default_rng(8)
    for rows, cols, zero in (([1], [1], False), ([5], [3], True), ([4, 9, 2], [3, 1, 2], False), ([33, 7], [32, 7], False)):
        rows = np.asarray(rows, np.int32); cols = np.asarray(cols, np.int32)
        n = int((rows.astype(np.int64) * cols).sum())
        tiles = np.zeros(n) if zero else rng.uniform(-1.0, 1.0, n)
        mat = qrkit_amd.SparseBlockDiagonal.fromTiles(rows, cols, tiles)
        qr = qrkit_amd.BlockDiagonalSparseQR(mat, context=ctx)
        _, ref = oracle_factorize(rows, cols, tiles)
        assert qr.info() == 0 and qr.rank() == ref.rank
        np.testing.assert_array_equal(qr.colsPermutation(), ref.perm)
        sq, sr, _ = tile_sizes(rows, cols)
        if zero:
            np.testing.assert_array_equal(qr.rValues().cpu().numpy(), ref.R_vals)
            np.testing.assert_array_equal(qr.qValues().cpu().numpy(), ref.Q_vals)
        else:
            assert per_tile_rel(qr.qValues().cpu().numpy(), ref.Q_vals, sq) <= RTOL
            assert per_tile_rel(qr.rValues().cpu().numpy(), ref.R_vals, sr) <= RTOL


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_random_mixed_batches_match_oracle(seed):
    """Randomised sweep over what a caller can hand in: mixed batches of 1..150 tiles with columns 1..70 and 0..9 extra rows
    (every kernel family and the bins between them in one matrix), values from one of four distributions (uniform, wide dynamic
    range, small integers = ties, one dominant column), both block solvers, both Q formats, optional trailing identity rows."""
    import qrkit_amd
    from qrkit_amd import _capi as capi
    rng = np.random.default_rng(1000 + seed)
    B = int(rng.integers(1, 151))
    cols = rng.integers(1, 71, B).astype(np.int32)
    rows = (cols + rng.integers(0, 10, B)).astype(np.int32)
    n = int((rows.astype(np.int64) * cols).sum())
    kind = seed % 4
    if kind == 0:
        tiles = rng.uniform(-1.0, 1.0, n)
    elif kind == 1:
        tiles = rng.uniform(-1.0, 1.0, n) * np.exp2(rng.integers(-30, 31, n))
    elif kind == 2:
        tiles = rng.integers(-2, 3, n).astype(np.float64)
    else:
        tiles = rng.uniform(-1.0, 1.0, n) * 1e-3
        off = 0
        for r, c in zip(rows, cols):
            tiles[off: off + r] += rng.uniform(1.0, 2.0, r)                                              # column 0 dominates
            off += int(r) * int(c)
    solver = capi.COLPIV_HOUSEHOLDER if seed % 3 else capi.HOUSEHOLDER
    qformat = capi.FULL_Q if seed % 2 else capi.BLOCK_DIAGONAL_Q
    extra = int(rng.integers(0, 4))
    mat_rows = int(rows.sum()) + extra
    mat = qrkit_amd.SparseBlockDiagonal.fromTiles(rows, cols, tiles, rows=mat_rows)
    qr = qrkit_amd.BlockDiagonalSparseQR(mat, blockSolver=solver, qFormat=qformat)
    _, ref = oracle_factorize(rows, cols, tiles, mat_rows=mat_rows, q_format=qformat, block_solver=solver)
    assert qr.info() == 0 and qr.rank() == ref.rank
    np.testing.assert_array_equal(qr.colsPermutation(), ref.perm)
    sq, sr, _ = tile_sizes(rows, cols)
    nq = int(sq.sum())
    assert per_tile_rel(qr.qValues().cpu().numpy()[:nq], ref.Q_vals[:nq], sq) <= 10 * RTOL
    assert per_tile_rel(qr.rValues().cpu().numpy(), ref.R_vals, sr) <= 10 * RTOL
    np.testing.assert_array_equal(qr.qValues().cpu().numpy()[nq:], ref.Q_vals[nq:])      # trailing identity rows


@pytest.mark.gpu
def test_mixed_8_to_256_full_share_of_one_gpu(qa, ctx):
    """BASELINE configs[4] at the count one of its 8 GPUs sees (100 000 / 8 = 12 500 square tiles, n ~ U{8..256}; three size
    classes on their streams, tiles handed out through the queues over several rounds of resident workgroups): the size-independent
    properties on every 25th tile, valid permutations everywhere, and a second factorisation that reproduces the first bit for bit."""
    import torch
    rng = np.random.default_rng(4)
    B = 12500
    n = rng.integers(8, 257, B).astype(np.int32)
    n64 = n.astype(np.int64)
    tiles = seeded_tiles(41, -1.0, 1.0, int((n64 ** 2).sum()))
    _, qr = run_gpu(qa, ctx, n, n, tiles)
    assert qr.info() == 0 and qr.rank() == int(n.sum())
    perm = qr.colsPermutation().copy()
    coff = np.concatenate([[0], np.cumsum(n64)]); toff = np.concatenate([[0], np.cumsum(n64 ** 2)])
    roff = np.concatenate([[0], np.cumsum(n64 * (n64 + 1) // 2)])
    # every permutation is a permutation of its own tile's columns
    local = perm - np.repeat(coff[:-1], n)
    assert local.min() >= 0 and np.all(local < np.repeat(n64, n))
    seen = np.zeros(int(coff[-1]), np.int32); seen[perm] += 1
    assert np.all(seen == 1)
    Qd, Rd = qr.qValues(), qr.rValues()
    worst_qr = worst_orth = 0.0
    for i in range(0, B, 25):
        c = int(n[i])
        A = tiles[toff[i]:toff[i + 1]].reshape(c, c).T
        Q = Qd[toff[i]:toff[i + 1]].cpu().numpy().reshape(c, c)
        li = np.tril_indices(c)
        R = np.zeros((c, c)); R[li[1], li[0]] = Rd[roff[i]:roff[i + 1]].cpu().numpy()
        P = perm[coff[i]:coff[i + 1]] - coff[i]
        worst_qr = max(worst_qr, np.linalg.norm(Q @ R - A[:, P]) / np.linalg.norm(A))
        worst_orth = max(worst_orth, np.linalg.norm(Q.T @ Q - np.eye(c)))
    assert worst_qr <= 1e-13 and worst_orth <= 1e-12, (worst_qr, worst_orth)
    q1, r1 = Qd.clone(), Rd.clone()
    _, qr2 = run_gpu(qa, ctx, n, n, tiles)
    np.testing.assert_array_equal(qr2.colsPermutation(), perm)
    assert torch.equal(qr2.qValues(), q1) and torch.equal(qr2.rValues(), r1)


@pytest.mark.parametrize("n,B", [(32, 3000), (8, 5000), (48, 700)])
def test_two_handles_on_two_streams_side_by_side(qa, n, B):
    """Independent matrices through two handles on their own HIP streams, launches alternating (what bench.py's concurrent_streams leg
    times): every factorisation must be bitwise what the same handle gives when it runs alone, and match the oracle."""
    import ctypes as C
    import torch
    from qrkit_amd import _capi as capi
    lib = capi.lib()
    dev = torch.device("cuda", 0)
    rows = np.full(B, n, np.int32)
    lanes = []
    for k in range(2):
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            c = qa.Context(0)
        lay = capi.BDLayout()
        lay.num_blocks, lay.block_rows, lay.block_cols = B, n, n
        lay.rows = lay.cols = None
        lay.mat_rows = lay.mat_cols = B * n
        plan = C.c_void_p()
        capi.check(lib.qrk_bd_plan_create(c.handle, C.byref(lay), capi.FULL_Q, capi.COLPIV_HOUSEHOLDER, C.byref(plan)), c.handle)
        g = torch.Generator(device=dev); g.manual_seed(100 + k)
        t = torch.rand(B * n * n, generator=g, device=dev, dtype=torch.float64) * 2 - 1
        q = torch.empty(B * n * n, device=dev, dtype=torch.float64)
        r = torch.empty(B * (n * (n + 1) // 2), device=dev, dtype=torch.float64)
        p = torch.empty(B * n, device=dev, dtype=torch.int32)
        lanes.append((st, c, plan, t, q, r, p))
    torch.cuda.synchronize()

    def fact(k):
        _, c, plan, t, q, r, p = lanes[k]
        capi.check(lib.qrk_bd_factorize(plan, t.data_ptr(), q.data_ptr(), r.data_ptr(), p.data_ptr(), None, capi.MEM_DEVICE), c.handle)
    alone = []
    for k in range(2):
        fact(k); torch.cuda.synchronize()
        alone.append((lanes[k][4].clone(), lanes[k][5].clone(), lanes[k][6].clone()))
        lanes[k][4].zero_(); lanes[k][5].zero_(); lanes[k][6].zero_()
    torch.cuda.synchronize()
    for it in range(6):
        fact(it % 2)
    torch.cuda.synchronize()
    for k in range(2):
        _, c, plan, t, q, r, p = lanes[k]
        assert torch.equal(q, alone[k][0]) and torch.equal(r, alone[k][1]) and torch.equal(p, alone[k][2]), "side by side differs from alone"
        _, ref = oracle_factorize(rows[:200], rows[:200], t[:200 * n * n].cpu().numpy())
        np.testing.assert_array_equal(p[:200 * n].cpu().numpy(), ref.perm)
        sq, sr, _ = tile_sizes(rows[:200], rows[:200])
        assert per_tile_rel(q[:200 * n * n].cpu().numpy(), ref.Q_vals, sq) <= RTOL
        assert per_tile_rel(r[:200 * (n * (n + 1) // 2)].cpu().numpy(), ref.R_vals, sr) <= RTOL
        lib.qrk_bd_plan_destroy(plan)


@pytest.mark.parametrize("B,r,c", [(1, 7, 2), (65, 8, 6), (300, 9, 2), (33, 16, 16), (7, 32, 32), (40, 20, 11), (5, 64, 40)])
def test_solve_r_alone_matches_per_tile_triangular_solves(qa, ctx, B, r, c):
    """qrk_bd_solve_r (the per-block back substitutions of BlockAngularSparseQR::makeR, BlockAngularSparseQR.h:285-335): z = R^-1 y per
    tile; the grouped kernel for tiles of at most 32 columns, the workgroup kernel above."""
    tiles = seeded_tiles(90 + B, 0.5, 5.0, B * r * c)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    _, qr = run_gpu(qa, ctx, rows, cols, tiles)
    Rv = qr.rValues().cpu().numpy().reshape(B, -1)
    y = np.random.default_rng(B).uniform(-1, 1, (B * c, 2))
    z = np.asarray(qr.solveR(y))
    iu = np.tril_indices(c)
    for i in (0, B // 2, B - 1):
        R = np.zeros((c, c)); R[iu[1], iu[0]] = Rv[i]
        want = np.linalg.solve(R, y[i * c:(i + 1) * c])
        assert np.linalg.norm(z[i * c:(i + 1) * c] - want) <= 1e-10 * max(1.0, np.linalg.norm(want))
