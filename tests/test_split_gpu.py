"""GPU parity of the opt-in two-kernel 32x32 path (bdqr_split.hip, QRK_SPLIT=1) against the oracle and the pair kernel."""
import os

import numpy as np
import pytest

from helpers import RTOL, oracle_factorize, rel_fro, seeded_tiles

pytestmark = pytest.mark.gpu


def run(flag, rows, cols, tiles):
    import qrkit_amd as qa
    if flag:
        os.environ["QRK_SPLIT"] = "1"
    try:
        qr = qa.BlockDiagonalSparseQR(context=qa.Context(0))
        qr.compute(qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles))
    finally:
        os.environ.pop("QRK_SPLIT", None)
    return qr


@pytest.mark.parametrize("B,lo,hi,seed", [(1000, 0.5, 5.0, 1), (257, -1.0, 1.0, 2), (5121, -1.0, 1.0, 3)])
def test_split_path_matches_oracle_and_pair_kernel(B, lo, hi, seed):
    r = c = 32
    tiles = seeded_tiles(seed, lo, hi, B * r * c)
    if seed == 2:                       # ties, zero columns, a zero tile, rank one: the rare branches
        t = tiles.reshape(B, c, r)
        t[0, 5] = t[0, 17]; t[1, 3] = 0.0; t[2] = 0.0; t[3] = np.outer(t[3, 0], np.ones(r)).T[:c]
        t[4] = np.eye(r)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    a, b = run(True, rows, cols, tiles), run(False, rows, cols, tiles)
    np.testing.assert_array_equal(a.colsPermutation(), b.colsPermutation())
    np.testing.assert_array_equal(a.rValues().cpu().numpy(), b.rValues().cpu().numpy())      # same operations, same order
    np.testing.assert_array_equal(a.qValues().cpu().numpy(), b.qValues().cpu().numpy())
    np.testing.assert_array_equal(a.hCoeffs().cpu().numpy(), b.hCoeffs().cpu().numpy())
    if seed != 2:
        _, ref = oracle_factorize(rows, cols, tiles)
        np.testing.assert_array_equal(a.colsPermutation(), ref.perm)
        assert rel_fro(a.rValues().cpu().numpy(), ref.R_vals) <= RTOL
        assert rel_fro(a.qValues().cpu().numpy(), ref.Q_vals) <= RTOL
