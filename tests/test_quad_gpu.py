"""GPU parity of the several-tiles-per-wavefront kernel (bdqr_quad.hip: uniform batches of tiles with 5..16 rows, cols <= rows; the design
of bdqr_pair4.hip at 16 rows, four tiles per wavefront, and at 8 rows, eight tiles per wavefront with two tiles per DPP row) against the
oracle, per tile; against bdqr_small.hip's lane groups, which it replaces for those shapes (QRK_QUAD=0); run to run; and at a large batch
by size-independent properties."""
import os

import numpy as np
import pytest

from helpers import oracle_factorize, per_tile_rel, seeded_tiles, tile_sizes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qa():
    import qrkit_amd
    return qrkit_amd


def factor(qa, rows, cols, tiles, solver=0, hc=True, quad=True):
    old = os.environ.get("QRK_QUAD")
    os.environ["QRK_QUAD"] = "1" if quad else "0"          # (read when the context is created)
    try:
        ctx = qa.Context(0)
        qr = qa.BlockDiagonalSparseQR(blockSolver=solver, qFormat=0, context=ctx, hCoeffs=hc)
        qr.compute(qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles))
    finally:
        if old is None:
            os.environ.pop("QRK_QUAD", None)
        else:
            os.environ["QRK_QUAD"] = old
    return qr


@pytest.mark.parametrize("B,r,c", [(1, 16, 16), (2, 16, 16), (3, 16, 16), (4, 16, 16), (5, 16, 16), (1001, 16, 16), (257, 9, 9), (130, 12, 12),
                                   (66, 16, 9), (67, 13, 7), (40, 10, 3), (33, 15, 15), (9, 11, 10), (12, 14, 14),
                                   # eight tiles per wavefront (5..8 rows): every count of tiles in the last wavefront
                                   (1, 8, 8), (2, 8, 8), (3, 8, 6), (7, 8, 8), (8, 8, 8), (9, 8, 8), (15, 7, 7), (1003, 8, 8), (777, 8, 6), (260, 6, 6),
                                   (131, 7, 4), (99, 5, 5), (64, 8, 3), (41, 6, 5), (23, 5, 3)])
@pytest.mark.parametrize("solver", [0, 1])
def test_quad_matches_oracle_per_tile(qa, B, r, c, solver):
    tiles = seeded_tiles(1000 + B + 16 * r + c, -1.0, 1.0, B * r * c)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    _, ref = oracle_factorize(rows, cols, tiles, block_solver=solver)
    nq, nr, nt = tile_sizes(rows, cols)
    for hc in (True, False):
        qr = factor(qa, rows, cols, tiles, solver, hc)
        assert qr.info() == ref.info and qr.rank() == ref.rank
        np.testing.assert_array_equal(qr.colsPermutation(), ref.perm)                       # bit-exact
        assert per_tile_rel(qr.qValues().cpu().numpy(), ref.Q_vals, nq) <= 1e-12
        assert per_tile_rel(qr.rValues().cpu().numpy(), ref.R_vals, nr) <= 1e-12
        if hc:
            assert per_tile_rel(qr.hCoeffs().cpu().numpy(), ref.hcoeffs, nt) <= 1e-12


def test_quad_kernel_is_the_one_that_runs(qa):
    import ctypes as C
    from qrkit_amd import _capi as capi
    capi.lib().qrk_bd_kernel_name.restype = C.c_char_p
    capi.lib().qrk_bd_kernel_name.argtypes = [C.c_void_p, C.c_int]
    for n in (16, 8, 5):
        rows = np.full(8, n, np.int32)
        qr = factor(qa, rows, rows, seeded_tiles(3, -1.0, 1.0, 8 * n * n))
        assert b"bdqr_quad_kernel" in capi.lib().qrk_bd_kernel_name(qr._plan, 0)


@pytest.mark.parametrize("r,c", [(16, 16), (12, 9), (9, 9), (8, 8), (8, 6), (6, 6), (5, 4)])
def test_quad_against_the_sixteen_lane_groups_and_run_to_run(qa, r, c):
    B = 403
    tiles = seeded_tiles(77 + r, 0.5, 5.0, B * r * c)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    a = factor(qa, rows, cols, tiles, quad=True)
    b = factor(qa, rows, cols, tiles, quad=True)
    s = factor(qa, rows, cols, tiles, quad=False)
    for x, y in ((a.qValues(), b.qValues()), (a.rValues(), b.rValues()), (a.hCoeffs(), b.hCoeffs())):
        assert np.array_equal(x.cpu().numpy(), y.cpu().numpy()), "two runs of the kernel differ bitwise"
    np.testing.assert_array_equal(a.colsPermutation(), s.colsPermutation())
    nq, nr, _ = tile_sizes(rows, cols)
    assert per_tile_rel(a.qValues().cpu().numpy(), s.qValues().cpu().numpy(), nq) <= 1e-12
    assert per_tile_rel(a.rValues().cpu().numpy(), s.rValues().cpu().numpy(), nr) <= 1e-12


@pytest.mark.parametrize("n", [16, 8])
@pytest.mark.parametrize("kind", ["pm1", "small_int", "dup_cols", "zero", "graded"])
def test_quad_tie_and_degenerate_tiles_take_the_exact_path(qa, kind, n):
    """Tiles whose decisions are ties or inside rounding: the permutation must be the oracle's, the values bitwise where the exact path ran."""
    rng = np.random.default_rng(5)
    B, r, c = 120, n, n
    if kind == "pm1":
        t = rng.choice([-1.0, 1.0], size=(B, c, r))
    elif kind == "small_int":
        t = rng.integers(-3, 4, size=(B, c, r)).astype(np.float64)
    elif kind == "dup_cols":
        t = rng.uniform(-1, 1, (B, c, r)); t[:, 5] = t[:, 2]; t[:, c - 1] = t[:, 2]
    elif kind == "zero":
        t = rng.uniform(-1, 1, (B, c, r)); t[::3] = 0.0; t[1::3, 4] = 0.0
    else:
        t = rng.uniform(-1, 1, (B, c, r)) * (2.0 ** (-3.0 * np.arange(c)))[None, :, None]
    tiles = t.reshape(-1)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    _, ref = oracle_factorize(rows, cols, tiles)
    qr = factor(qa, rows, cols, tiles)
    np.testing.assert_array_equal(qr.colsPermutation(), ref.perm)
    nq, nr, _ = tile_sizes(rows, cols)
    if kind == "pm1":
        assert np.array_equal(qr.rValues().cpu().numpy(), ref.R_vals), "every tile of this kind goes through the exact path: bitwise the oracle"
        assert np.array_equal(qr.qValues().cpu().numpy(), ref.Q_vals)
    elif kind in ("small_int", "graded"):
        assert per_tile_rel(qr.qValues().cpu().numpy(), ref.Q_vals, nq) <= 1e-12
        assert per_tile_rel(qr.rValues().cpu().numpy(), ref.R_vals, nr) <= 1e-12
    else:
        # rank-deficient tiles: Q is not unique beyond the rank; A P = Q R and Q^T Q = I per tile
        Q = qr.qValues().cpu().numpy().reshape(B, r, r)
        Rv = qr.rValues().cpu().numpy().reshape(B, -1)
        P = qr.colsPermutation().reshape(B, c) - (np.arange(B) * c)[:, None]
        iu = np.tril_indices(c)
        for i in range(B):
            Rm = np.zeros((r, c)); Rm[iu[1], iu[0]] = Rv[i]
            A = t[i].T
            assert np.linalg.norm(Q[i] @ Rm - A[:, P[i]]) <= 1e-12 * max(1.0, np.linalg.norm(A))
            assert np.linalg.norm(Q[i].T @ Q[i] - np.eye(r)) <= 1e-12


@pytest.mark.parametrize("n", [16, 8])
def test_quad_large_batch_by_properties(qa, n):
    import torch
    B = 200000
    g = torch.Generator(device="cuda").manual_seed(9)
    tiles = torch.rand(B * n * n, device="cuda", dtype=torch.float64, generator=g) * 2 - 1
    rows = np.full(B, n, np.int32)
    qr = factor(qa, rows, rows, tiles, hc=False)
    Q = qr.qValues().reshape(B, n, n)
    Rv = qr.rValues().reshape(B, -1)
    iu = torch.tril_indices(n, n, device="cuda")
    R = torch.zeros((B, n, n), device="cuda", dtype=torch.float64)
    R[:, iu[1], iu[0]] = Rv
    perm = torch.as_tensor(np.asarray(qr.colsPermutation()), device="cuda").long().reshape(B, n) - (torch.arange(B, device="cuda") * n)[:, None]
    assert bool((torch.sort(perm, dim=1).values == torch.arange(n, device="cuda")[None, :]).all())
    A = tiles.reshape(B, n, n).transpose(1, 2)                    # tiles are column-major
    AP = torch.gather(A, 2, perm[:, None, :].expand(B, n, n))
    err = (torch.linalg.matrix_norm(Q @ R - AP) / torch.linalg.matrix_norm(A)).max().item()
    orth = torch.linalg.matrix_norm(Q.transpose(1, 2) @ Q - torch.eye(n, device="cuda", dtype=torch.float64)).max().item()
    d = torch.diagonal(R, dim1=1, dim2=2).abs()
    assert err <= 1e-13 and orth <= 1e-13
    assert bool((d[:, :-1] >= d[:, 1:] * (1 - 1e-12)).all()), "|R_kk| must not increase"
