"""The one-wave-per-tile on-chip kernel for tiles with 32 < rows <= 64 (qrkit_amd/csrc/bdqr_w64.hip) against the CPU oracle, through
the C ABI: every boundary of its layout -- the 64-row frame (tiles aligned to its bottom: 33, 34, 47, 48, 49, 63, 64 rows), the
16-row chunks of the DPP broadcast (columns that end on and next to a chunk boundary), the narrowest tiles of the class (1 and 2
columns), rectangular tiles -- with both block solvers, with and without tau, uniform and mixed launches (the queue hands a wave
tiles of different shapes one after the other), tiles that must go through the exact path, and bdqr_col.hip's LDS-resident form
(QRK_W64=0) as a cross-check."""
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
    (40, 33, 33),     # narrowest square tile of the class: 31 steps of the unrolled sequence skipped
    (9, 34, 33),
    (9, 33, 1),       # one reflector
    (9, 40, 2),
    (9, 47, 47),
    (9, 48, 48),      # exactly three chunks
    (9, 49, 48),
    (9, 49, 17),      # the last column ends one past a chunk boundary
    (9, 63, 63),
    (9, 64, 63),
    (9, 64, 1),
    (9, 64, 16),
    (9, 64, 33),
    (600, 64, 64),    # the full frame; more tiles than the launch has waves: the queue
    (9, 50, 32),      # cols <= 32 < rows
    # more tiles than resident waves at the heights where the LDS of the launch changes the workgroups per CU (12 up to 52 rows,
    # 11, 10, 9, then 8 from 61 rows on): every wave slot of a CU in use, the queue behind them
    (3200, 33, 33),
    (3200, 52, 52),
    (3000, 53, 40),
    (2800, 57, 57),
    (2600, 61, 30),
]


@pytest.mark.parametrize("B,r,c", SHAPES)
@pytest.mark.parametrize("solver", [0, 1])
@pytest.mark.parametrize("hc", [True, False])
def test_uniform_tiles_match_oracle(qa, ctx, B, r, c, solver, hc):
    tiles = seeded_tiles(r * 1000 + c, -1.0, 1.0, B * r * c)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    _, qr = run_gpu(qa, ctx, rows, cols, tiles, solver=solver, hc=hc)
    _, ref = oracle_factorize(rows, cols, tiles, block_solver=solver)
    compare(qr, ref, rows, cols, hc=hc)


def test_kernel_name_of_the_class(qa, ctx):
    """The plan routes the class to the new kernel (and says so): no silent fallback to the LDS-resident form."""
    import ctypes as C
    from qrkit_amd import _capi as capi
    lay = capi.BDLayout()
    lay.num_blocks, lay.block_rows, lay.block_cols = 4, 64, 64
    lay.rows = lay.cols = None
    lay.mat_rows = lay.mat_cols = 256
    plan = C.c_void_p()
    capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), 0, 0, C.byref(plan)), ctx.handle)
    capi.lib().qrk_bd_kernel_name.restype = C.c_char_p
    capi.lib().qrk_bd_kernel_name.argtypes = [C.c_void_p, C.c_int]
    assert capi.lib().qrk_bd_kernel_name(plan, 0).decode() == "qrk::bdqr_w64_kernel<true>"
    capi.lib().qrk_bd_plan_destroy(plan)


def test_mixed_launch_every_alignment(qa, ctx):
    """One launch whose tiles start at many rows of the frame, in random order, next to tiles of the other kernel families, both Q
    formats, trailing identity rows."""
    rng = np.random.default_rng(78)
    rows = np.concatenate([np.arange(33, 65), [64, 64, 40, 33, 50, 20, 8, 100, 200, 32]]).astype(np.int32)
    cols = np.minimum(rows, np.concatenate([np.arange(33, 65) - rng.integers(0, 20, 32), [64, 10, 40, 33, 32, 20, 6, 80, 130, 32]])).astype(np.int32)
    order = rng.permutation(len(rows))
    rows, cols = rows[order], cols[order]
    n = int((rows.astype(np.int64) * cols).sum())
    tiles = seeded_tiles(6, -1.0, 1.0, n)
    for solver in (0, 1):
        for qf in (0, 1):
            _, qr = run_gpu(qa, ctx, rows, cols, tiles, mat_rows=int(rows.sum()) + 3, q_format=qf, solver=solver)
            _, ref = oracle_factorize(rows, cols, tiles, mat_rows=int(rows.sum()) + 3, q_format=qf, block_solver=solver)
            compare(qr, ref, rows, cols)


@pytest.mark.parametrize("r,c", [(64, 64), (48, 40), (33, 33)])
def test_decisions_inside_the_margin_go_to_the_exact_path(qa, ctx, r, c):
    """Ties, zero and duplicated columns, a rank-one tile, entries +-1: the kernel flags the tile and bdqr_exact.hip redoes it in
    Eigen's own operation order -- permutation, R and Q are the oracle's, bit for bit."""
    rng = np.random.default_rng(9)
    t = []
    a = rng.uniform(-1, 1, (r, c)); a[:, 5] = a[:, c - 3]; a[:, 20] = a[:, c - 3]; t.append(a)    # duplicate columns
    a = rng.uniform(-1, 1, (r, c)); a[:, 3] = 0.0; a[:, c - 1] = 0.0; t.append(a)                  # zero columns
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
    r, c, B = 64, 60, 12
    rng = np.random.default_rng(4)
    t = []
    for b in range(B):
        a = rng.uniform(-1, 1, (r, c))
        if b < 8:
            a *= np.exp2(rng.integers(-30, 31, c))[None, :]
        else:
            a *= 1e-3; a[:, 0] += rng.uniform(1.0, 2.0, r)
        t.append(a)
    tiles = np.concatenate([x.ravel(order="F") for x in t])
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    _, qr = run_gpu(qa, ctx, rows, cols, tiles)
    _, ref = oracle_factorize(rows, cols, tiles)
    compare(qr, ref, rows, cols, tol=10 * RTOL)


def test_properties_at_scale(qa, ctx):
    """20 000 tiles of 64 x 64: A P = Q R, Q^T Q = I, the diagonal of R non-increasing, and two runs bitwise equal."""
    import torch
    B, r = 20000, 64
    g = torch.Generator(device="cuda").manual_seed(3)
    tiles = torch.rand(B * r * r, device="cuda", dtype=torch.float64, generator=g) * 2 - 1
    rows = np.full(B, r, np.int32)
    mat = qa.SparseBlockDiagonal.fromTiles(rows, rows, tiles)
    qr = qa.BlockDiagonalSparseQR(context=ctx)
    qr.compute(mat)
    Q = qr.qValues().reshape(B, r, r)
    Rp = qr.rValues().reshape(B, r * (r + 1) // 2)
    P = torch.as_tensor(qr.colsPermutation(), device="cuda").reshape(B, r) - (torch.arange(B, device="cuda") * r)[:, None]
    il = torch.tril_indices(r, r, device="cuda")            # packed by columns = row-major order of the transposed lower triangle
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
    assert bool((P.sort(dim=1).values == torch.arange(r, device="cuda")[None, :]).all())
    q1, r1 = qr.qValues().clone(), qr.rValues().clone()
    qr.factorize(mat)
    assert torch.equal(q1, qr.qValues()) and torch.equal(r1, qr.rValues())


def test_lds_resident_form_agrees():
    """QRK_W64=0 selects bdqr_col.hip's LDS-resident form for the same tiles (kept for comparison): same permutation, Q and R within
    the tolerance of the fast path."""
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from helpers import RTOL, oracle_factorize, per_tile_rel, seeded_tiles, tile_sizes
import qrkit_amd as qa
rows = np.array([64, 33, 48, 50, 64], np.int32); cols = np.array([64, 33, 40, 32, 10], np.int32)
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
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, QRK_W64="0"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("seed", range(6))
def test_random_batches_match_oracle(seed):
    """Randomised sweep over the class: 20...60 tiles with 33...64 rows and 1...rows columns, values from one of four distributions
    (uniform, wide dynamic range, small integers = ties everywhere -> exact path, one dominant column), both block solvers, both Q
    formats."""
    import qrkit_amd
    from qrkit_amd import _capi as capi
    rng = np.random.default_rng(9000 + seed)
    B = int(rng.integers(20, 61))
    rows = rng.integers(33, 65, B).astype(np.int32)
    cols = np.minimum(rows, rng.integers(1, 65, B)).astype(np.int32)
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
