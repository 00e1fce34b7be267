"""GPU: the HIP path against the committed golden fixtures (tests/golden/*.npz)."""
import glob
import os

import numpy as np
import pytest

from helpers import RTOL, rel_fro

pytestmark = pytest.mark.gpu
ALL_GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
GOLDEN = [p for p in ALL_GOLDEN if "kind" not in np.load(p).files]          # block-diagonal fixtures
OTHER = [p for p in ALL_GOLDEN if "kind" in np.load(p).files]               # dense / banded / angular


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


@pytest.mark.parametrize("path", OTHER, ids=[os.path.basename(p)[:-4] for p in OTHER])
def test_hip_matches_golden_compositions(path):
    """The dense right-block solver and the two compositions against their committed fixtures."""
    import scipy.sparse as sp
    import torch
    import qrkit_amd
    g = np.load(path)
    kind = str(g["kind"])
    if kind == "dense":
        from qrkit_amd.angular import DenseColPivQR
        A = g["A"]
        qr = DenseColPivQR(qrkit_amd.Context(0), 0)
        At = torch.from_numpy(np.asfortranarray(A).T.copy()).cuda().t()
        qr.compute(At)
        np.testing.assert_array_equal(qr.colsPermutation().cpu().numpy(), g["perm"])
        assert rel_fro(At.cpu().numpy(), g["packed"]) <= 1e-11
        assert rel_fro(qr._hc.cpu().numpy(), g["hcoeffs"]) <= 1e-11
    elif kind == "banded":
        J = sp.csr_matrix((g["data"], g["indices"], g["indptr"]), shape=tuple(g["shape"]))
        qr = qrkit_amd.BandedBlockedSparseQR(suggestedBlockCols=int(g["suggested"]))
        qr.compute(J)
        np.testing.assert_array_equal(qr.blocks, g["blocks"])
        np.testing.assert_array_equal(qr.rowsPermutation(), g["row_perm"])
        R = qr.matrixR()
        np.testing.assert_array_equal(R.indptr, g["r_indptr"])
        np.testing.assert_array_equal(R.indices, g["r_indices"])
        Rg = sp.csc_matrix((g["r_data"], g["r_indices"], g["r_indptr"]), shape=R.shape)
        # rows whose reflector has an exactly-zero leading entry carry a noise-determined sign (see test_banded.py)
        flip = np.where(np.sign(R.diagonal()) != np.sign(Rg.diagonal()))[0]
        assert len(flip) <= len(g["blocks"])
        D = np.ones(R.shape[0]); D[flip] = -1.0
        assert rel_fro((sp.diags(D) @ R).toarray(), Rg.toarray()) <= 1e-12
    else:
        left = qrkit_amd.SparseBlockDiagonal.fromTiles(g["rows"], g["cols"], g["tiles"])
        ba = qrkit_amd.BlockAngularSparseQR()
        ba.compute(qrkit_amd.BlockMatrix1x2(left, g["J2"]))
        np.testing.assert_array_equal(ba.colsPermutation(), g["perm"])
        assert ba.rank() == int(g["rank"])
        Rg = sp.csc_matrix((g["r_data"], g["r_indices"], g["r_indptr"]), shape=ba.matrixR().shape)
        assert rel_fro(ba.matrixR().toarray(), Rg.toarray()) <= 1e-12
