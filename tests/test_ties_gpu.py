"""GPU tie battery: inputs whose pivot decisions are ties in exact arithmetic (sign, indicator, small-integer, circulant,
repeated-column tiles) or sit on the edge of Eigen's tests.  Eigen -- and the oracle, which restates its scalar operation
order -- breaks such ties by the rounding noise of its own arithmetic, so the permutation is only reproducible by doing that
arithmetic: the fast kernels flag every decision that is not clear of rounding and the exact path (bdqr_exact.hip) redoes the
tile.  Bar: column permutation bit-exact on EVERY tile; Q, R, tau within 1e-12 per tile -- and bit-identical on the tiles the
exact path produced (all of them, when QRK_EXACT=1)."""
import os

import numpy as np
import pytest

from helpers import RTOL, oracle_factorize, per_tile_rel, seeded_tiles, tile_sizes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qa():
    import qrkit_amd
    return qrkit_amd


@pytest.fixture(scope="module")
def ctx(qa):
    return qa.Context(0)


@pytest.fixture(scope="module")
def ctx_exact(qa):
    """A context whose every factorisation runs the exact path (QRK_EXACT is read by qrk_create)."""
    os.environ["QRK_EXACT"] = "1"
    try:
        return qa.Context(0)
    finally:
        os.environ.pop("QRK_EXACT", None)


def tie_tiles(kind, B, r, c, seed):
    """B column-major tiles r x c of one tie family, packed."""
    rng = np.random.default_rng(seed)
    if kind == "pm1":                       # +-1: every column norm equal, every dot product a small integer
        a = rng.choice([-1.0, 1.0], size=(B, c, r))
    elif kind == "zero_one":                # indicator columns
        a = rng.integers(0, 2, size=(B, c, r)).astype(np.float64)
    elif kind == "small_int":               # integers -3..3
        a = rng.integers(-3, 4, size=(B, c, r)).astype(np.float64)
    elif kind == "pm1_pow2":                # +-1 columns scaled by powers of two: ties reappear after the first steps
        a = rng.choice([-1.0, 1.0], size=(B, c, r)) * np.exp2(rng.integers(-3, 4, size=(B, c, 1)))
    elif kind == "pow2_scaled":             # generic columns scaled over 40 binades (no ties: the margins must not fire wrongly)
        a = rng.uniform(-1, 1, size=(B, c, r)) * np.exp2(rng.integers(-20, 21, size=(B, c, 1)))
    elif kind == "circulant":               # columns are rotations of one vector: norms equal up to summation order
        v = rng.uniform(-1, 1, size=(B, r))
        a = np.stack([np.roll(v, s, axis=1) for s in range(c)], axis=1)
    elif kind == "dup_cols":                # repeated and negated columns
        a = rng.uniform(-1, 1, size=(B, c, r))
        for b in range(B):
            src = rng.integers(0, c, size=c // 3)
            dst = rng.integers(0, c, size=c // 3)
            a[b, dst] = a[b, src] * rng.choice([-1.0, 1.0], size=(c // 3, 1))
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(a).reshape(-1)      # [B][c][r] = column-major tiles back to back


KINDS = ["pm1", "zero_one", "small_int", "pm1_pow2", "pow2_scaled", "circulant", "dup_cols"]


def check(qr, ref, rows, cols, bitwise=False):
    assert qr.info() == ref.info and qr.rank() == ref.rank
    np.testing.assert_array_equal(qr.colsPermutation(), ref.perm)
    sq, sr, sc = tile_sizes(rows, cols)
    Q, R, tau = qr.qValues().cpu().numpy(), qr.rValues().cpu().numpy(), qr.hCoeffs().cpu().numpy()
    if bitwise:
        np.testing.assert_array_equal(R, ref.R_vals)
        np.testing.assert_array_equal(tau, ref.hcoeffs)
        np.testing.assert_array_equal(Q, ref.Q_vals)
        return
    # rank-deficient tiles: what follows the collapse is rounding noise of the particular operation order, on both sides;
    # such tiles meet a tie or an edge of a test on the way and are redone by the exact path, i.e. they are bit-identical.
    # So every tile is either bit-identical or a clear-cut tile within the tolerance.
    nq = int(sq.sum())
    eq = lambda a, b, sizes: np.array([np.array_equal(x, y) for x, y in zip(np.split(a, np.cumsum(sizes)[:-1]),
                                                                          np.split(b, np.cumsum(sizes)[:-1]))])
    same = eq(R, ref.R_vals, sr) & eq(Q[:nq], ref.Q_vals[:nq], sq)
    idx = np.flatnonzero(~same)
    if idx.size:
        sel = lambda a, sizes: np.concatenate([np.split(a, np.cumsum(sizes)[:-1])[i] for i in idx])
        assert per_tile_rel(sel(R, sr), sel(ref.R_vals, sr), sr[idx]) <= RTOL
        assert per_tile_rel(sel(Q[:nq], sq), sel(ref.Q_vals[:nq], sq), sq[idx]) <= RTOL
        assert per_tile_rel(sel(tau, sc), sel(ref.hcoeffs, sc), sc[idx]) <= RTOL
    return int(same.sum())


# 32x32: K1 (uniform persistent kernel); 24x17 / 20x20: K1's ragged kernel; 7x2, 8x6, 9x2, 16x16: K5; 64x64, 48x40, 33x33: K2, one wave per
# tile (bdqr_w64.hip); 100x48: K2, LDS-resident (bdqr_col.hip); 300x40: the workgroup kernel for tiles above 256
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("B,r,c", [(300, 32, 32), (300, 24, 17), (300, 7, 2), (300, 8, 6), (300, 16, 16), (64, 64, 64),
                                   (48, 48, 40), (64, 33, 33), (24, 100, 48), (6, 300, 40)])
def test_tie_battery_permutation_bit_exact(qa, ctx, kind, B, r, c):
    tiles = tie_tiles(kind, B, r, c, seed=KINDS.index(kind) * 1000 + 37 * r + c)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    mat = qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles)
    qr = qa.BlockDiagonalSparseQR(context=ctx)
    qr.compute(mat)
    _, ref = oracle_factorize(rows, cols, tiles)
    check(qr, ref, rows, cols)


@pytest.mark.parametrize("kind", ["pm1", "circulant", "small_int"])
def test_tie_battery_mixed_sizes(qa, ctx, kind):
    """A mixed batch (every kernel class in one factorize) of tie tiles."""
    rng = np.random.default_rng(3)
    B = 150
    cols = rng.integers(2, 70, B).astype(np.int32)
    rows = (cols + rng.integers(0, 5, B)).astype(np.int32)
    parts = [tie_tiles(kind, 1, int(r), int(c), seed=int(100 + i)) for i, (r, c) in enumerate(zip(rows, cols))]
    tiles = np.concatenate(parts)
    mat = qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles)
    qr = qa.BlockDiagonalSparseQR(context=ctx)
    qr.compute(mat)
    _, ref = oracle_factorize(rows, cols, tiles)
    check(qr, ref, rows, cols)


@pytest.mark.parametrize("solver", [0, 1])
@pytest.mark.parametrize("B,r,c,lo,hi,seed", [
    (200, 32, 32, -1.0, 1.0, 2), (256, 7, 2, 0.5, 5.0, 1), (100, 8, 6, -1.0, 1.0, 3), (20, 64, 64, -1.0, 1.0, 1),
    (6, 100, 37, -1.0, 1.0, 2), (10, 1, 1, -1.0, 1.0, 6), (3, 300, 40, -1.0, 1.0, 7),
])
def test_exact_path_is_bitwise_the_oracle(qa, ctx_exact, B, r, c, lo, hi, seed, solver):
    """QRK_EXACT=1: every tile through bdqr_exact.hip -- permutation, tau, R and Q bit-identical to the oracle
    (sequential IEEE operations in Eigen's order; correctly rounded / and sqrt on the device)."""
    tiles = seeded_tiles(seed, lo, hi, B * r * c)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    mat = qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles)
    qr = qa.BlockDiagonalSparseQR(blockSolver=solver, context=ctx_exact)
    qr.compute(mat)
    _, ref = oracle_factorize(rows, cols, tiles, block_solver=solver)
    check(qr, ref, rows, cols, bitwise=True)


@pytest.mark.parametrize("kind", ["pm1", "zero_one", "circulant"])
def test_exact_path_bitwise_on_tie_tiles(qa, ctx_exact, kind):
    B, r, c = 100, 32, 32
    tiles = tie_tiles(kind, B, r, c, seed=11)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    mat = qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles)
    qr = qa.BlockDiagonalSparseQR(context=ctx_exact)
    qr.compute(mat)
    _, ref = oracle_factorize(rows, cols, tiles)
    check(qr, ref, rows, cols, bitwise=True)


def test_exact_path_mixed_sizes_bitwise(qa, ctx_exact):
    rng = np.random.default_rng(12345)
    B = 60
    n = rng.integers(8, 130, B).astype(np.int32)
    tiles = seeded_tiles(21, -1.0, 1.0, int((n.astype(np.int64) ** 2).sum()))
    mat = qa.SparseBlockDiagonal.fromTiles(n, n, tiles, rows=int(n.sum()) + 3)
    qr = qa.BlockDiagonalSparseQR(context=ctx_exact)
    qr.compute(mat)
    _, ref = oracle_factorize(n, n, tiles, mat_rows=int(n.sum()) + 3)
    check(qr, ref, n, n, bitwise=True)


def test_generic_tiles_are_not_flagged(qa, ctx):
    """The margins must not send generic data to the slow path: on uniform random 32x32 tiles the fast kernel's results
    (FMA chains, not bit-identical to the oracle) survive, i.e. no tile was redone."""
    B, r, c = 2000, 32, 32
    tiles = seeded_tiles(9, -1.0, 1.0, B * r * c)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    mat = qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles)
    qr = qa.BlockDiagonalSparseQR(context=ctx)
    qr.compute(mat)
    _, ref = oracle_factorize(rows, cols, tiles)
    n_same = check(qr, ref, rows, cols)
    assert n_same <= 2, f"{n_same} of {B} generic tiles took the exact path"


@pytest.mark.parametrize("r,c", [(8, 8), (12, 7), (20, 17), (31, 31)])
def test_generic_small_and_ragged_tiles_stay_on_the_fast_path(qa, ctx, r, c):
    """The same for the small-tile kernel (G = 8, 16) and the ragged form of the pair kernel.  With so few operations a tile can
    agree with the oracle bit for bit by chance, so the bar is that most tiles do not (a false-positive flag redoes ALL of them)."""
    B = 256
    rng = np.random.default_rng(100 * r + c)
    tiles = rng.uniform(0.5, 5.0, B * r * c)
    rows = np.full(B, r, np.int32); cols = np.full(B, c, np.int32)
    qr = qa.BlockDiagonalSparseQR(qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles), context=ctx)
    _, ref = oracle_factorize(rows, cols, tiles)
    np.testing.assert_array_equal(qr.colsPermutation(), ref.perm)
    got = qr.qValues().cpu().numpy()
    per = r * r
    redone = sum(np.array_equal(got[i * per:(i + 1) * per], ref.Q_vals[i * per:(i + 1) * per]) for i in range(B))
    assert redone < B // 2, f"{redone} of {B} generic {r}x{c} tiles are bitwise the exact path's result"


@pytest.mark.parametrize("n", [16, 32, 33, 48, 64, 96, 160, 300])
def test_generic_tiles_stay_on_the_fast_path(qa, ctx, n):
    """The other half of the decision margins: on generic data NO tile may be flagged.  The exact path rounds like a scalar
    evaluation of Eigen's algorithm and is bit-identical to the oracle; the fast kernels (FMA chains, squared norms) differ from it
    in the last bits -- so a tile whose R is bitwise the oracle's went through the exact path.  (A false positive is invisible to
    the parity tests, the exact result being right: round 2 had every tile of 33..64 columns redone, 5x slower, because threads
    of already chosen columns read their stale |x_tail|^2 = 0 as a degenerate reflector.)  One kernel family per size class."""
    B = 64 if n <= 160 else 8
    rng = np.random.default_rng(n)
    tiles = rng.uniform(0.5, 5.0, B * n * n)
    rows = np.full(B, n, np.int32)
    mat = qa.SparseBlockDiagonal.fromTiles(rows, rows, tiles)
    qr = qa.BlockDiagonalSparseQR(mat, context=ctx)
    _, ref = oracle_factorize(rows, rows, tiles)
    np.testing.assert_array_equal(qr.colsPermutation(), ref.perm)
    got = qr.rValues().cpu().numpy()
    per = n * (n + 1) // 2
    redone = sum(np.array_equal(got[i * per:(i + 1) * per], ref.R_vals[i * per:(i + 1) * per]) for i in range(B))
    assert redone == 0, f"{redone} of {B} generic {n}x{n} tiles were sent to the exact path"
