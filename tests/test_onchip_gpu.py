"""The on-chip kernel for tiles wider than 64 columns (qrkit_amd/csrc/bdqr_reg.hip: registers + LDS, 512 threads per tile) against
the CPU oracle, through the C ABI: every boundary of its layout -- the padded frame (rows = 256), the LDS rows (tiles taller than
192 rows), each of the six register-chunk levels (a level dies every 32 rows), the 16-row panels of the Q accumulation (cols and
rows that are not multiples of 16 / 4), the narrowest tile of the class (65 columns), the boundary between the 4-wave (at most 128 columns and 192 rows, two workgroups per
CU) and the 8-wave instantiation -- with both block solvers, uniform and mixed
launches, tiles that must go through the exact path, and the old global-workspace form (QRK_COL_ONCHIP=0) as a cross-check."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import RTOL, oracle_factorize, per_tile_rel, seeded_tiles, tile_sizes
from test_bd_gpu import compare, run_gpu

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def qa():
    import qrkit_amd
    return qrkit_amd


@pytest.fixture(scope="module")
def ctx(qa):
    return qa.Context(0)


SHAPES = [
    (3, 65, 65),      # narrowest tile of the class; starts with four register levels dead
    (3, 66, 65),
    (2, 97, 96),      # rows not a multiple of 4
    (2, 128, 100),
    (3, 128, 128),    # the widest tile of the 4-wave instantiation (two workgroups per CU)
    (2, 192, 128),    # ... and its tallest
    (2, 193, 128),    # one row more: the 8-wave instantiation
    (2, 140, 129),    # one column more
    (2, 160, 160),    # exactly three live levels
    (2, 161, 70),
    (2, 191, 191),
    (2, 192, 192),    # the tallest tile without LDS rows
    (2, 193, 130),    # one LDS row
    (2, 224, 224),
    (2, 255, 255),
    (2, 256, 65),     # tall and narrow: 191 rows below the last reflector
    (2, 256, 200),
    (2, 256, 256),
]


@pytest.mark.parametrize("B,r,c", SHAPES)
@pytest.mark.parametrize("solver", [0, 1])
def test_uniform_tiles_match_oracle(qa, ctx, B, r, c, solver):
    tiles = seeded_tiles(r * 1000 + c, -1.0, 1.0, B * r * c)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    _, qr = run_gpu(qa, ctx, rows, cols, tiles, solver=solver)
    _, ref = oracle_factorize(rows, cols, tiles, block_solver=solver)
    compare(qr, ref, rows, cols)


def test_mixed_launch_all_levels(qa, ctx):
    """One launch whose tiles start at every level (the queue hands a workgroup tiles of different sizes one after the other:
    nothing of a tile may survive into the next), next to tiles of the other kernel families."""
    rng = np.random.default_rng(77)
    rows = np.array([256, 70, 200, 129, 96, 255, 65, 180, 33, 16, 224, 100, 64, 150, 256, 90], np.int32)
    cols = np.array([256, 66, 150, 129, 80, 200, 65, 170, 33, 16, 224, 100, 64, 70, 90, 90], np.int32)
    order = rng.permutation(len(rows))
    rows, cols = rows[order], cols[order]
    n = int((rows.astype(np.int64) * cols).sum())
    tiles = seeded_tiles(5, -1.0, 1.0, n)
    for solver in (0, 1):
        _, qr = run_gpu(qa, ctx, rows, cols, tiles, solver=solver)
        _, ref = oracle_factorize(rows, cols, tiles, block_solver=solver)
        compare(qr, ref, rows, cols)


def test_decisions_inside_the_margin_go_to_the_exact_path(qa, ctx):
    """Ties, zero and duplicated columns, a rank-one tile, entries +-1: the on-chip kernel flags the tile and bdqr_exact.hip redoes
    it in Eigen's own operation order -- permutation, R and Q are the oracle's, bit for bit."""
    r, c = 200, 120
    rng = np.random.default_rng(9)
    t = []
    a = rng.uniform(-1, 1, (r, c)); a[:, 5] = a[:, 77]; a[:, 100] = a[:, 77]; t.append(a)        # duplicate columns
    a = rng.uniform(-1, 1, (r, c)); a[:, 3] = 0.0; a[:, 119] = 0.0; t.append(a)                   # zero columns
    t.append(np.zeros((r, c)))                                                                    # all zero
    t.append(np.outer(rng.uniform(-1, 1, r), rng.uniform(-1, 1, c)))                              # rank one
    t.append(rng.choice([-1.0, 1.0], (r, c)))                                                     # all column norms equal: a tie at step 0
    tiles = np.concatenate([x.ravel(order="F") for x in t])
    B = len(t)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    _, qr = run_gpu(qa, ctx, rows, cols, tiles)
    _, ref = oracle_factorize(rows, cols, tiles)
    np.testing.assert_array_equal(qr.colsPermutation(), ref.perm)
    np.testing.assert_array_equal(qr.rValues().cpu().numpy(), ref.R_vals)
    np.testing.assert_array_equal(qr.qValues().cpu().numpy(), ref.Q_vals)


def test_wide_dynamic_range_and_graded_columns(qa, ctx):
    """Columns scaled over 60 binary orders of magnitude (the norm downdate recomputes often) and a tile with one dominant column."""
    r, c, B = 256, 160, 3
    rng = np.random.default_rng(4)
    t = []
    for b in range(B):
        a = rng.uniform(-1, 1, (r, c))
        if b < 2:
            a *= np.exp2(rng.integers(-30, 31, c))[None, :]
        else:
            a *= 1e-3; a[:, 0] += rng.uniform(1.0, 2.0, r)
        t.append(a)
    tiles = np.concatenate([x.ravel(order="F") for x in t])
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    _, qr = run_gpu(qa, ctx, rows, cols, tiles)
    _, ref = oracle_factorize(rows, cols, tiles)
    compare(qr, ref, rows, cols, tol=10 * RTOL)


def test_properties_at_full_share(qa, ctx):
    """600 tiles of 256 x 256 (more than two per CU: the queue, the reuse of the workspace frame): A P = Q R, Q^T Q = I, the
    diagonal of R non-increasing, and two runs bitwise equal."""
    import torch
    B, r = 600, 256
    g = torch.Generator(device="cuda").manual_seed(3)
    tiles = torch.rand(B * r * r, device="cuda", dtype=torch.float64, generator=g) * 2 - 1
    rows = np.full(B, r, np.int32)
    mat = qa.SparseBlockDiagonal.fromTiles(rows, rows, tiles)
    qr = qa.BlockDiagonalSparseQR(context=ctx)
    qr.compute(mat)
    Q = qr.qValues().reshape(B, r, r)
    Rp = qr.rValues().reshape(B, r * (r + 1) // 2)
    P = torch.as_tensor(qr.colsPermutation(), device="cuda").reshape(B, r) - (torch.arange(B, device="cuda") * r)[:, None]
    iu = torch.triu_indices(r, r, device="cuda")            # packed by columns = row-major order of the transposed lower triangle
    il = torch.tril_indices(r, r, device="cuda")
    R = torch.zeros(B, r, r, device="cuda", dtype=torch.float64)
    R[:, il[1], il[0]] = Rp
    A = tiles.reshape(B, r, r).transpose(1, 2)              # tiles are column-major
    AP = torch.gather(A, 2, P[:, None, :].expand(B, r, r))
    err = (torch.bmm(Q, R) - AP).flatten(1).norm(dim=1) / AP.flatten(1).norm(dim=1)
    assert float(err.max()) <= 1e-13
    orth = (torch.bmm(Q.transpose(1, 2), Q) - torch.eye(r, device="cuda", dtype=torch.float64)).flatten(1).norm(dim=1)
    assert float(orth.max()) <= 1e-12
    d = R.diagonal(dim1=1, dim2=2).abs()
    assert bool((d[:, 1:] <= d[:, :-1] * (1 + 1e-12)).all())
    assert sorted(P[0].tolist()) == list(range(r)) and sorted(P[-1].tolist()) == list(range(r))
    q1, r1 = qr.qValues().clone(), qr.rValues().clone()
    qr.factorize(mat)
    assert torch.equal(q1, qr.qValues()) and torch.equal(r1, qr.rValues())
    del iu


def test_old_global_workspace_form_agrees():
    """QRK_COL_ONCHIP=0 selects bdqr_col.hip's global-workspace form for the same tiles (kept for comparison): same permutation, Q and
    R within the tolerance of the fast path."""
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from helpers import RTOL, oracle_factorize, per_tile_rel, seeded_tiles, tile_sizes
import qrkit_amd as qa
rows = np.array([256, 100, 192, 70], np.int32); cols = np.array([200, 100, 192, 66], np.int32)
tiles = seeded_tiles(11, -1.0, 1.0, int((rows.astype(np.int64) * cols).sum()))
mat = qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles)
qr = qa.BlockDiagonalSparseQR(mat)
_, ref = oracle_factorize(rows, cols, tiles)
np.testing.assert_array_equal(qr.colsPermutation(), ref.perm)
sq, sr, _ = tile_sizes(rows, cols)
assert per_tile_rel(qr.rValues().cpu().numpy(), ref.R_vals, sr) <= RTOL
assert per_tile_rel(qr.qValues().cpu().numpy(), ref.Q_vals, sq) <= RTOL
print("OK")
''' % (ROOT, os.path.join(ROOT, "tests"))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, QRK_COL_ONCHIP="0"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("seed", range(8))
def test_random_large_tile_batches_match_oracle(seed):
    """Randomised sweep over the on-chip class: 4...14 tiles with 65...256 columns and 0...60 extra rows (capped at 256), mixed with a few
    small ones, values from one of four distributions (uniform, wide dynamic range, small integers = ties everywhere -> exact path, one
    dominant column), both block solvers, both Q formats."""
    import qrkit_amd
    from qrkit_amd import _capi as capi
    rng = np.random.default_rng(7000 + seed)
    B = int(rng.integers(4, 15))
    cols = rng.integers(65, 257, B).astype(np.int32)
    rows = np.minimum(256, cols + rng.integers(0, 61, B)).astype(np.int32)
    small = rng.integers(1, 65, 3).astype(np.int32)
    cols = np.concatenate([cols, small]); rows = np.concatenate([rows, small + rng.integers(0, 5, 3).astype(np.int32)])
    order = rng.permutation(len(cols)); cols, rows = cols[order], rows[order]
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
            tiles[off: off + r] += rng.uniform(1.0, 2.0, r)
            off += int(r) * int(c)
    solver = capi.COLPIV_HOUSEHOLDER if seed % 3 else capi.HOUSEHOLDER
    qformat = capi.FULL_Q if seed % 2 else capi.BLOCK_DIAGONAL_Q
    mat = qrkit_amd.SparseBlockDiagonal.fromTiles(rows, cols, tiles)
    qr = qrkit_amd.BlockDiagonalSparseQR(mat, blockSolver=solver, qFormat=qformat)
    _, ref = oracle_factorize(rows, cols, tiles, q_format=qformat, block_solver=solver)
    assert qr.info() == 0 and qr.rank() == ref.rank
    np.testing.assert_array_equal(qr.colsPermutation(), ref.perm)
    sq, sr, _ = tile_sizes(rows, cols)
    assert per_tile_rel(qr.qValues().cpu().numpy(), ref.Q_vals, sq) <= 10 * RTOL
    assert per_tile_rel(qr.rValues().cpu().numpy(), ref.R_vals, sr) <= 10 * RTOL
