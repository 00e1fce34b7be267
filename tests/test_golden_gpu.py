"""GPU: the HIP path against the committed golden fixtures (tests/golden/*.npz)."""
import glob
import os

import numpy as np
import pytest

from helpers import RTOL, rel_fro

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_hip_matches_golden(path):
    import qrkit_amd
    g = np.load(path)
    mat = qrkit_amd.SparseBlockDiagonal.fromTiles(g["rows"], g["cols"], g["tiles"], rows=int(g["mat_rows"]))
    qr = qrkit_amd.BlockDiagonalSparseQR(blockSolver=int(g["solver"]), qFormat=int(g["q_format"]))
    qr.compute(mat)
    assert qr.info() == 0 and qr.rank() == int(g["rank"])
    np.testing.assert_array_equal(qr.colsPermutation(), g["perm"])
    assert rel_fro(qr.qValues().cpu().numpy(), g["Q_vals"]) <= RTOL
    assert rel_fro(qr.rValues().cpu().numpy(), g["R_vals"]) <= RTOL
    assert rel_fro(qr.hCoeffs().cpu().numpy(), g["hcoeffs"]) <= RTOL
    for got, key in zip(qr.pattern(), ("q_rowptr", "q_colidx", "r_colptr", "r_rowidx")):
        np.testing.assert_array_equal(got, g[key])
