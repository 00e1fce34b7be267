"""GPU: the persistent form of the pivoted dense QR (qrkit_amd/csrc/dense_qr_pers.hip: up to 2048 x 2048 in the registers of one launch over
the whole chip, an XCD-hierarchical grid barrier per reflector that also elects the pivot) against the oracle -- direct calls and as
the second stage of the two-stage form -- and against the launch-per-reflector form (the default: the persistent form measured slower
and is opt-in, QRK_DENSE_PERS=1)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import rel_fro
from oracle import oracle as orc
from test_dense_gpu import _factor

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _persistent_form(monkeypatch):
    monkeypatch.setenv("QRK_DENSE_PERS", "1")         # opt-in: the launch per reflector is the faster form (profiles/r04_k3_stage2.txt)


@pytest.mark.parametrize("rows,cols", [(256, 256), (300, 300), (1000, 257), (2048, 260), (1300, 1300), (2048, 2048)])
@pytest.mark.parametrize("solver", [0, 1])
def test_persistent_dense_qr_matches_oracle(rows, cols, solver):
    rng = np.random.default_rng(rows * 5 + cols)
    A = rng.uniform(-1.0, 1.0, (rows, cols)) * rng.uniform(0.5, 2.0, cols)[None, :]
    qr, At = _factor(A, solver, "cols")
    got = At.cpu().numpy()
    if solver == 0:
        ref, hc, perm, _ = orc.colpiv_qr(A)
        np.testing.assert_array_equal(qr.colsPermutation().cpu().numpy(), perm)      # bit-exact
    else:
        ref, hc = orc.householder_qr(A)
        np.testing.assert_array_equal(qr.colsPermutation().cpu().numpy(), np.arange(cols))
    assert rel_fro(np.triu(got[:cols]), np.triu(ref[:cols])) <= 1e-11
    row_err = np.linalg.norm(np.triu(got[:cols]) - np.triu(ref[:cols]), axis=1) / np.linalg.norm(np.triu(ref[:cols]), axis=1)
    assert row_err.max() <= 1e-10
    assert rel_fro(qr._hc.cpu().numpy()[:cols], hc[:cols]) <= 1e-11
    assert rel_fro(np.tril(got, -1), np.tril(ref, -1)) <= 1e-10


@pytest.mark.parametrize("kind", ["pm1", "dup_cols"])
def test_persistent_tie_matrix_goes_to_the_exact_path(kind):
    rng = np.random.default_rng(3)
    rows, cols = 400, 300
    if kind == "pm1":
        A = rng.choice([-1.0, 1.0], (rows, cols))
    else:
        A = rng.uniform(-1, 1, (rows, cols)); A[:, 7] = A[:, 200]; A[:, 100] = A[:, 200]
    qr, At = _factor(A, 0, "cols")
    ref, hc, perm, _ = orc.colpiv_qr(A)
    np.testing.assert_array_equal(qr.colsPermutation().cpu().numpy(), perm)
    np.testing.assert_array_equal(At.cpu().numpy(), ref)                             # the exact path's result is the oracle's, bit for bit


def test_launch_per_reflector_form_agrees():
    """QRK_DENSE_PERS=0 keeps dense_qr_cols.hip's launch per reflector for the same matrix: same permutation, R within rounding."""
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from test_dense_gpu import _factor
rng = np.random.default_rng(5)
A = rng.uniform(-1, 1, (700, 400))
qr, At = _factor(A, 0, "cols")
np.save(sys.argv[1], At.cpu().numpy()); np.save(sys.argv[2], qr.colsPermutation().cpu().numpy())
''' % (ROOT, os.path.join(ROOT, "tests"))
    import tempfile
    d = tempfile.mkdtemp()
    outs = []
    for flag in ("1", "0"):
        f1, f2 = os.path.join(d, f"a{flag}.npy"), os.path.join(d, f"p{flag}.npy")
        out = subprocess.run([sys.executable, "-c", code, f1, f2], env=dict(os.environ, QRK_DENSE_PERS=flag), capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        outs.append((np.load(f1), np.load(f2)))
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    assert rel_fro(np.triu(outs[0][0][:400]), np.triu(outs[1][0][:400])) <= 1e-12


def test_second_stage_of_the_two_stage_form(monkeypatch):
    """6000 x 300: CAQR, then the 300 x 300 triangle through the persistent kernel: permutation bit-exact, rows of R up to sign."""
    monkeypatch.setenv("QRK_DENSE_TWO_STAGE", "1")
    rng = np.random.default_rng(8)
    rows, cols = 6000, 300
    A = rng.uniform(-1.0, 1.0, (rows, cols)) * rng.uniform(0.5, 2.0, cols)[None, :]
    qr, At = _factor(A, 0, None)
    ref, hc, perm, _ = orc.colpiv_qr(A)
    np.testing.assert_array_equal(qr.colsPermutation().cpu().numpy(), perm)
    Rg, Rr = np.triu(At.cpu().numpy()[:cols]), np.triu(ref[:cols])
    sg = np.sign(np.diag(Rg)) * np.sign(np.diag(Rr))
    row_err = np.linalg.norm(Rg * sg[:, None] - Rr, axis=1) / np.linalg.norm(Rr, axis=1)
    assert row_err.max() <= 1e-11
