#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run in the build container, NOT on the GPU box).

The reference (jasvob/QRKit) cannot be built or imported here: it is header-only C++ over Eigen >= 3.3,
which is absent from this image and from /root/reference, and its own tests hold no golden numbers
(only invariants, test/test-qrkit.cpp:201-203).  So the fixtures pin VALUES as follows:
  inputs   : the reference's deterministic generator (libstdc++ default_random_engine +
             uniform_real_distribution, test/test-qrkit.cpp:64-65,101-117) restated in oracle/qrk_oracle.c,
             plus seeded U(-1,1) tiles from the same engine;
  outputs  : the CPU oracle (restatement of Eigen's ColPivHouseholderQR / HouseholderQR /
             HouseholderSequence and of BlockDiagonalSparseQR::factorize);
  confirmed: against LAPACK dgeqp3 / dgeqrf via SciPy at generation time (pivots identical, R/Q within
             1e-13) -- LAPACK uses the same reflector convention and LAWN-176 pivot rule.
The fixtures are data (inputs + expected outputs) only.
"""
import os
import sys

import numpy as np
import scipy.linalg as sl

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as orc  # noqa: E402


def lapack_confirm(r, c, tiles, res, solver):
    B = len(tiles) // (r * c)
    qoff = roff = 0
    for i in range(B):
        A = tiles[i * r * c:(i + 1) * r * c].reshape(c, r).T
        Q = res.Q_vals[qoff:qoff + r * r].reshape(r, r)
        li = np.tril_indices(c)
        R = np.zeros((c, c)); R[li[1], li[0]] = res.R_vals[roff:roff + c * (c + 1) // 2]
        P = res.perm[i * c:(i + 1) * c] - i * c
        if solver == orc.COLPIV:
            Qs, Rs, Ps = sl.qr(A, pivoting=True)
            assert np.array_equal(Ps, P), f"tile {i}: pivots differ from LAPACK dgeqp3"
        else:
            Qs, Rs = sl.qr(A)
            assert np.array_equal(P, np.arange(c))
        scale = max(np.abs(Rs).max(), 1e-300)
        assert np.abs(Rs[:c] - R).max() <= 1e-13 * scale, f"tile {i}: R differs from LAPACK"
        assert np.abs(Qs - Q).max() <= 1e-12, f"tile {i}: Q differs from LAPACK"
        qoff += r * r; roff += c * (c + 1) // 2


def make_case(name, B, r, c, tiles, solver=orc.COLPIV, q_format=orc.FULL_Q, extra_rows=0):
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    prob = orc.BDProblem(rows, cols, tiles, matRows=B * r + extra_rows, q_format=q_format, block_solver=solver)
    res = prob.factorize()
    assert res.info == 0
    lapack_confirm(r, c, tiles, res, solver)
    qp, qi, rp, ri = prob.pattern()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), rows=rows, cols=cols, tiles=tiles, mat_rows=B * r + extra_rows,
                        solver=solver, q_format=q_format, Q_vals=res.Q_vals, R_vals=res.R_vals, perm=res.perm,
                        hcoeffs=res.hcoeffs, rank=res.rank, q_rowptr=qp, q_colidx=qi, r_colptr=rp, r_rowidx=ri)
    print(f"{name}: {B} tiles {r}x{c} ok")


def main():
    # the reference's own test input: generate_block_diagonal_matrix, first 32 of the 256 7x2 blocks
    make_case("ref_7x2_colpiv", 32, 7, 2, orc.gen_reference_7x2(32))
    make_case("ref_7x2_colpiv_bdq", 32, 7, 2, orc.gen_reference_7x2(32), q_format=orc.BLOCK_DIAGONAL_Q, extra_rows=3)
    make_case("u05_32x32_colpiv", 6, 32, 32, orc.gen_uniform(1, 0.5, 5.0, 6 * 1024))
    make_case("u11_32x32_colpiv", 6, 32, 32, orc.gen_uniform(2, -1.0, 1.0, 6 * 1024))
    make_case("u11_32x32_nopiv", 4, 32, 32, orc.gen_uniform(3, -1.0, 1.0, 4 * 1024), solver=orc.NOPIV)
    make_case("u11_8x6_colpiv", 16, 8, 6, orc.gen_uniform(4, -1.0, 1.0, 16 * 48))
    make_case("u11_6x6_colpiv", 16, 6, 6, orc.gen_uniform(5, -1.0, 1.0, 16 * 36))
    make_case("u05_2x1_colpiv", 32, 2, 1, orc.gen_uniform(6, 0.5, 5.0, 32 * 2))
    make_case("u11_30x20_colpiv", 4, 30, 20, orc.gen_uniform(7, -1.0, 1.0, 4 * 600))


def make_compositions():
    """Mid-size tiles (bdqr_col.hip), the dense right-block solver and the two compositions: inputs from the
    reference's generators / seeded uniforms, expected outputs from the oracle."""
    import scipy.sparse as sp
    sys.path.insert(0, os.path.dirname(HERE))
    from test_banded import banded_matrix
    from test_angular import angular_problem
    make_case("u11_64x64_colpiv", 2, 64, 64, orc.gen_uniform(8, -1.0, 1.0, 2 * 4096))
    make_case("u11_100x37_colpiv", 2, 100, 37, orc.gen_uniform(9, -1.0, 1.0, 2 * 3700))
    # dense ColPivHouseholderQR with implicit Q (the right-block solver)
    A = orc.gen_uniform(10, -1.0, 1.0, 300 * 40).reshape(40, 300).T.copy()
    qr, hc, perm, _ = orc.colpiv_qr(A)
    Qs, Rs, Ps = sl.qr(A, pivoting=True)
    assert np.array_equal(Ps, perm) and np.abs(np.triu(qr[:40]) - Rs[:40]).max() <= 1e-12 * np.abs(Rs).max()
    np.savez_compressed(os.path.join(HERE, "dense_300x40_colpiv.npz"), kind="dense", A=A, packed=qr, hcoeffs=hc, perm=perm)
    print("dense_300x40_colpiv ok")
    # banded: the reference's overlapping input (test-qrkit.cpp:62-128) at 32 variables, SuggestedBlockCols = 8
    J = banded_matrix(32, True, None)
    res = orc.bb_factorize(J, 8)
    R = sp.csc_matrix(res.R); R.sort_indices()
    np.savez_compressed(os.path.join(HERE, "banded_overlap_32.npz"), kind="banded", indptr=J.indptr, indices=J.indices,
                        data=J.data, shape=np.array(J.shape), suggested=8, blocks=np.array(res.blocks, dtype=np.int32),
                        row_perm=res.row_perm, r_indptr=R.indptr, r_indices=R.indices, r_data=R.data)
    print("banded_overlap_32 ok:", len(res.blocks), "blocks")
    # angular: 64 tiles of 7x2 + 24 dense columns
    prob, tiles, J1, J2 = angular_problem(64, 24)
    ref = orc.ba_factorize(prob, J2)
    R = sp.csc_matrix(ref.R); R.sort_indices()
    np.savez_compressed(os.path.join(HERE, "angular_64x7x2_24.npz"), kind="angular", rows=prob.rows, cols=prob.cols,
                        tiles=tiles, J2=J2, perm=ref.perm, rank=ref.rank, r_indptr=R.indptr, r_indices=R.indices,
                        r_data=R.data)
    print("angular_64x7x2_24 ok")


def make_small():
    """Shapes of the small-tile kernel (bdqr_small.hip): one per lane-group size G = 4, 8, 16, the LM-damped 9x2 blocks
    of the reference (rowpermADiagLambda, test/test-utils.cpp:145-180, on its own 7x2 input, lambda = 1e-3) and a
    HouseholderQR (no pivoting) case."""
    make_case("u11_4x4_colpiv", 24, 4, 4, orc.gen_uniform(11, -1.0, 1.0, 24 * 16))
    make_case("u11_3x2_colpiv", 24, 3, 2, orc.gen_uniform(12, -1.0, 1.0, 24 * 6))
    make_case("u11_8x8_colpiv", 12, 8, 8, orc.gen_uniform(13, -1.0, 1.0, 12 * 64))
    make_case("u11_16x16_colpiv", 6, 16, 16, orc.gen_uniform(14, -1.0, 1.0, 6 * 256))
    make_case("u11_12x7_colpiv", 8, 12, 7, orc.gen_uniform(15, -1.0, 1.0, 8 * 84))
    make_case("u11_16x5_nopiv", 8, 16, 5, orc.gen_uniform(16, -1.0, 1.0, 8 * 80), solver=orc.NOPIV)
    nv = 24
    j7 = orc.gen_reference_7x2(nv).reshape(nv, 2, 7)
    damped = np.zeros((nv, 2, 9))
    damped[:, :, :7] = j7
    damped[:, 0, 7] = np.sqrt(1e-3)
    damped[:, 1, 8] = np.sqrt(1e-3)
    make_case("ref_9x2_lm_damped_colpiv", nv, 9, 2, damped.ravel())


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "small":
        make_small()
    else:
        main()
        make_compositions()
        make_small()
