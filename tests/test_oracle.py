"""CPU tests of the oracle: golden fixtures, LAPACK cross-check, the reference's invariants, generator."""
import glob
import os
import subprocess
import tempfile

import numpy as np
import pytest
import scipy.linalg as sl

from helpers import rel_fro
from oracle import oracle as orc

ALL_GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
GOLDEN = [p for p in ALL_GOLDEN if "kind" not in np.load(p).files]          # block-diagonal fixtures
OTHER = [p for p in ALL_GOLDEN if "kind" in np.load(p).files]               # dense / banded / angular


@pytest.mark.parametrize("path", OTHER, ids=[os.path.basename(p)[:-4] for p in OTHER])
def test_oracle_reproduces_golden_compositions(path):
    import scipy.sparse as sp
    g = np.load(path)
    kind = str(g["kind"])
    if kind == "dense":
        qr, hc, perm, _ = orc.colpiv_qr(g["A"])
        np.testing.assert_array_equal(perm, g["perm"])
        np.testing.assert_array_equal(qr, g["packed"])
        np.testing.assert_array_equal(hc, g["hcoeffs"])
    elif kind == "banded":
        J = sp.csr_matrix((g["data"], g["indices"], g["indptr"]), shape=tuple(g["shape"]))
        res = orc.bb_factorize(J, int(g["suggested"]))
        np.testing.assert_array_equal(np.array(res.blocks, dtype=np.int32), g["blocks"])
        np.testing.assert_array_equal(res.row_perm, g["row_perm"])
        R = sp.csc_matrix(res.R); R.sort_indices()
        np.testing.assert_array_equal(R.indptr, g["r_indptr"])
        np.testing.assert_array_equal(R.indices, g["r_indices"])
        np.testing.assert_array_equal(R.data, g["r_data"])
    else:
        prob = orc.BDProblem(g["rows"], g["cols"], g["tiles"])
        ref = orc.ba_factorize(prob, g["J2"])
        np.testing.assert_array_equal(ref.perm, g["perm"])
        assert ref.rank == int(g["rank"])
        R = sp.csc_matrix(ref.R); R.sort_indices()
        np.testing.assert_array_equal(R.indptr, g["r_indptr"])
        np.testing.assert_array_equal(R.data, g["r_data"])


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    g = np.load(path)
    prob = orc.BDProblem(g["rows"], g["cols"], g["tiles"], matRows=int(g["mat_rows"]), q_format=int(g["q_format"]),
                         block_solver=int(g["solver"]))
    res = prob.factorize()
    assert res.info == 0 and res.rank == int(g["rank"])
    np.testing.assert_array_equal(res.perm, g["perm"])
    np.testing.assert_array_equal(res.Q_vals, g["Q_vals"])   # same code, same flags: bitwise
    np.testing.assert_array_equal(res.R_vals, g["R_vals"])
    np.testing.assert_array_equal(res.hcoeffs, g["hcoeffs"])
    for got, key in zip(prob.pattern(), ("q_rowptr", "q_colidx", "r_colptr", "r_rowidx")):
        np.testing.assert_array_equal(got, g[key])
    # the "faithful assembly" variant (per-element insertion + triplet sort) gives the same matrices
    res2 = prob.factorize(faithful=True)
    np.testing.assert_array_equal(res2.perm, res.perm)
    np.testing.assert_array_equal(res2.Q_vals, res.Q_vals)
    np.testing.assert_array_equal(res2.R_vals, res.R_vals)


def test_survey_anchor_7x2():
    """SURVEY.md Appendix C anchor: block 0 of the reference's generated input and its pivoted QR."""
    t = orc.gen_reference_7x2(1)
    assert t[0] == 1.0919200448244228 and t[7] == 2.8836508691347573 and t[13] == 3.4426353298123522
    qr, hc, perm, _ = orc.colpiv_qr(t.reshape(2, 7).T)
    np.testing.assert_array_equal(perm, [1, 0])
    np.testing.assert_allclose(np.triu(qr)[:2], [[-8.012310461458963, -6.356917986189552], [0, -3.581484054054724]], rtol=1e-15)
    np.testing.assert_allclose(hc, [1.359902537851694, 1.6138312476594023], rtol=1e-15)


def test_generator_matches_libstdcxx():
    """The C restatement of minstd_rand0 + generate_canonical + uniform_real_distribution vs libstdc++ itself."""
    src = r"""
#include <random>
#include <cstdio>
int main(){ std::default_random_engine gen; std::uniform_real_distribution<double> dist(0.5,5.0);
  for(int i=0;i<2000;i++) std::printf("%a\n", dist(gen)); return 0; }
"""
    with tempfile.TemporaryDirectory() as d:
        cpp, exe = os.path.join(d, "g.cpp"), os.path.join(d, "g")
        open(cpp, "w").write(src)
        subprocess.check_call(["g++", "-O1", cpp, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    want = np.array([float.fromhex(x) for x in out])
    got = orc.gen_uniform(1, 0.5, 5.0, 2000)
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("m,n,lo,hi,count", [(7, 2, 0.5, 5, 300), (32, 32, -1, 1, 60), (6, 6, -1, 1, 200), (8, 6, -1, 1, 200),
                                             (64, 64, -1, 1, 6), (200, 37, -1, 1, 4), (2, 1, 0.5, 5, 100)])
def test_colpiv_matches_lapack(m, n, lo, hi, count):
    rng = np.random.default_rng(m * 100 + n)
    for _ in range(count):
        A = rng.uniform(lo, hi, (m, n))
        qr, hc, perm, _ = orc.colpiv_qr(A)
        Q, R = orc.form_q(qr, hc), np.triu(qr)[:n]
        Qs, Rs, Ps = sl.qr(A, pivoting=True)
        np.testing.assert_array_equal(Ps, perm)
        assert np.abs(Rs[:n] - R).max() <= 1e-13 * np.abs(R).max()
        assert np.abs(Qs - Q).max() <= 1e-12


def test_householder_and_block_factor_match_lapack():
    rng = np.random.default_rng(3)
    for m, n in [(7, 4), (32, 32), (448, 192), (14, 4), (9, 2)]:
        A = rng.uniform(-1, 1, (m, n))
        qr, hc = orc.householder_qr(A)
        (qr2, tau2), _ = sl.qr(A, mode="raw")
        assert np.abs(qr - qr2).max() <= 1e-13 and np.abs(hc - tau2).max() <= 1e-13
        V = np.tril(qr, -1)[:, :n] + np.eye(m, n)
        T = orc.block_triangular_factor(V, hc)
        assert np.abs(np.triu(T) - T).max() == 0
        assert np.abs(np.eye(m) - V @ T @ V.T - orc.form_q(qr, hc)).max() <= 1e-13   # H_0..H_{n-1} = I - V T V^T


def test_reference_invariants_on_reference_input():
    """test_block_diagonal (test/test-qrkit.cpp:167-206) on the oracle: Q R = J P, Q^T J P = R, LS recovery,
    at 1e-12 (the reference's bar is 1e-6)."""
    import scipy.sparse as sp
    nv = 256
    tiles = orc.gen_reference_7x2(nv)
    prob = orc.BDProblem.uniform(nv, 7, 2, tiles)
    res = prob.factorize()
    qp, qi, rp, ri = prob.pattern()
    Q = sp.csr_matrix((res.Q_vals, qi, qp), shape=(7 * nv, 7 * nv))
    R = sp.csc_matrix((res.R_vals, ri, rp), shape=(7 * nv, 2 * nv))
    J = sp.block_diag([tiles[i * 14:(i + 1) * 14].reshape(2, 7).T for i in range(nv)], format="csc")
    JP = J[:, res.perm]
    assert rel_fro((Q @ R).toarray(), JP.toarray()) <= 1e-12
    assert rel_fro((Q.T @ JP).toarray(), R.toarray()) <= 1e-12
    x = np.random.default_rng(0).uniform(-1, 1, 2 * nv)
    assert rel_fro(prob.solve(res, J @ x), x) <= 1e-12


def test_rank_deficient_and_ties():
    r = c = 8
    A = np.ones((r, c))
    qr, hc, perm, nz = orc.colpiv_qr(A)
    assert perm[0] == 0 and nz == 1          # first maximum wins the tie
    Z = np.zeros((r, c))
    qr, hc, perm, nz = orc.colpiv_qr(Z)
    np.testing.assert_array_equal(perm, np.arange(c))
    assert np.all(hc == 0)      # (Eigen keeps nonzero_pivots = size here: 0 < 0 is false)
    Q = orc.form_q(qr, hc)
    np.testing.assert_array_equal(Q, np.eye(r))


def test_landscape_tile_is_invalid_input():
    prob = orc.BDProblem([3], [5], np.arange(15.0))
    assert prob.factorize().info == orc.INVALID_INPUT
