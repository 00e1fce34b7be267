"""GPU parity of the small-tile kernel (bdqr_small.hip: 64/G tiles per wavefront, uniform batches with rows <= 16)
against the oracle and against the pair kernel it replaces for those shapes (QRK_SMALL=0)."""
import os

import numpy as np
import pytest

from helpers import RTOL, oracle_factorize, rel_fro, seeded_tiles

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qa():
    import qrkit_amd
    return qrkit_amd


@pytest.fixture(scope="module")
def ctx(qa):
    return qa.Context(0)


def gpu(qa, ctx, rows, cols, tiles, solver=0, q_format=0):
    qr = qa.BlockDiagonalSparseQR(blockSolver=solver, qFormat=q_format, context=ctx)
    qr.compute(qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles))
    return qr


@pytest.mark.parametrize("B,r,c", [
    (1000, 7, 2), (1000, 9, 2),                 # the reference's test shape and its LM-damped form
    (777, 6, 6), (777, 8, 6),                   # BASELINE block-angular left part
    (5, 4, 4), (63, 3, 2), (64, 4, 1), (65, 2, 2),      # G = 4, batch sizes around one workgroup (64 tiles)
    (31, 8, 8), (33, 5, 5), (129, 8, 1),        # G = 8 (32 tiles per workgroup)
    (16, 16, 16), (17, 16, 5), (100, 12, 12), (15, 9, 9), (50, 13, 7),   # G = 16 (16 tiles per workgroup)
    (1, 1, 1), (3, 16, 1),
])
@pytest.mark.parametrize("solver", [0, 1])
def test_small_uniform_tiles_match_oracle(qa, ctx, B, r, c, solver):
    tiles = seeded_tiles(B + r, -1.0, 1.0, B * r * c)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    for qf in (0, 1):
        qr = gpu(qa, ctx, rows, cols, tiles, solver, qf)
        _, ref = oracle_factorize(rows, cols, tiles, q_format=qf, block_solver=solver)
        assert qr.info() == ref.info and qr.rank() == ref.rank
        np.testing.assert_array_equal(qr.colsPermutation(), ref.perm)        # bit-exact
        assert rel_fro(qr.rValues().cpu().numpy(), ref.R_vals) <= RTOL
        assert rel_fro(qr.qValues().cpu().numpy(), ref.Q_vals) <= RTOL
        assert rel_fro(qr.hCoeffs().cpu().numpy(), ref.hcoeffs) <= RTOL


@pytest.mark.parametrize("r,c", [(8, 8), (16, 12), (4, 4), (7, 2)])
def test_small_ties_zero_and_rank_deficient_tiles(qa, ctx, r, c):
    """Exact ties resolve to the first maximum (smallest current position), zero tiles take Eigen's tau = 0 branch."""
    rng = np.random.default_rng(r * 100 + c)
    t = []
    a = rng.uniform(-1, 1, (r, c)); a[:, c - 1] = a[:, 0]; t.append(a)                         # duplicate columns
    a = rng.uniform(-1, 1, (r, c)); a[:, 0] = 0.0; t.append(a)                                # a zero column
    t.append(np.eye(r)[:, :c].copy())                                                         # all norms equal
    t.append(np.zeros((r, c)))                                                                # all zero
    t.append(np.outer(rng.uniform(-1, 1, r), rng.uniform(-1, 1, c)))                          # rank one
    t.append(np.ones((r, c)))
    a = np.eye(r)[:, :c].copy(); a[:, [0, c - 1]] = a[:, [c - 1, 0]]; t.append(a)             # ties after swaps
    tiles = np.concatenate([x.ravel(order="F") for x in t])
    B = len(t)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    qr = gpu(qa, ctx, rows, cols, tiles)
    _, ref = oracle_factorize(rows, cols, tiles)
    P = qr.colsPermutation()
    for i in (2, 3, 5, 6):          # structural ties are decided by the first-maximum rule alone: bit-exact
        np.testing.assert_array_equal(P[i * c:(i + 1) * c], ref.perm[i * c:(i + 1) * c])
    Qv = qr.qValues().cpu().numpy().reshape(B, r, r)
    Rv = qr.rValues().cpu().numpy().reshape(B, -1)
    for i in range(B):
        li = np.tril_indices(c); Rm = np.zeros((r, c)); Rm[li[1], li[0]] = Rv[i]   # packed upper triangle by columns
        Pi = P[i * c:(i + 1) * c] - i * c
        assert sorted(Pi.tolist()) == list(range(c))
        assert np.linalg.norm(Qv[i] @ Rm - t[i][:, Pi]) <= 1e-13 * max(1.0, np.linalg.norm(t[i]))
        assert np.linalg.norm(Qv[i].T @ Qv[i] - np.eye(r)) <= 1e-13
        d = np.abs(np.diag(Rm[:c, :c]))
        assert np.all(d[:-1] >= d[1:] * (1 - 1e-12) - 1e-300)          # non-increasing |R_kk|


def test_small_kernel_agrees_with_pair_kernel(qa):
    """Same batch through bdqr_small.hip and (QRK_SMALL=0) through bdqr_pair.hip: identical permutation, Q/R to rounding."""
    B, r, c = 3000, 8, 6
    tiles = seeded_tiles(9, 0.5, 5.0, B * r * c)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    out = []
    for flag in ("1", "0"):
        os.environ["QRK_SMALL"] = flag
        try:
            qr = gpu(qa, qa.Context(0), rows, cols, tiles)
        finally:
            os.environ.pop("QRK_SMALL", None)
        out.append((qr.colsPermutation().copy(), qr.qValues().cpu().numpy(), qr.rValues().cpu().numpy()))
    np.testing.assert_array_equal(out[0][0], out[1][0])
    assert rel_fro(out[0][1], out[1][1]) <= 1e-14 and rel_fro(out[0][2], out[1][2]) <= 1e-14


def test_small_tiles_properties_at_baseline_size(qa, ctx):
    """BASELINE block-angular left part: 20000 tiles of 6x6 and of 8x6; A P = Q R, Q^T Q = I on every tile."""
    import torch
    for r, c in ((6, 6), (8, 6)):
        B = 20000
        tiles = seeded_tiles(4, -1.0, 1.0, B * r * c)
        rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
        qr = gpu(qa, ctx, rows, cols, tiles)
        Q = qr.qValues().reshape(B, r, r)
        Rp = qr.rValues().reshape(B, -1)
        li = np.tril_indices(c)
        R = torch.zeros(B, r, c, dtype=torch.float64, device=Q.device)
        R[:, torch.as_tensor(li[1]), torch.as_tensor(li[0])] = Rp
        P = torch.as_tensor(qr.colsPermutation().reshape(B, c) % c, device=Q.device).long()
        A = torch.as_tensor(tiles.reshape(B, c, r), device=Q.device).transpose(1, 2)
        AP = torch.gather(A, 2, P[:, None, :].expand(B, r, c))
        assert float((Q @ R - AP).norm() / AP.norm()) <= 1e-14
        eye = torch.eye(r, dtype=torch.float64, device=Q.device)
        assert float((Q.transpose(1, 2) @ Q - eye).abs().max()) <= 1e-14
        assert sorted(P[0].tolist()) == list(range(c))
