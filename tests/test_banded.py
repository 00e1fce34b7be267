"""Block-banded composition (BandedBlockedSparseQR): the oracle against the reference's invariants on the
reference's own three inputs (CPU), and the HIP path against the oracle (GPU)."""
import numpy as np
import pytest
import scipy.sparse as sp

from helpers import rel_fro
from oracle import oracle as orc


def banded_matrix(num_vars, overlap, shuffle_seed=None, seed=1):
    """generate_block_diagonal_matrix / generate_overlapping_block_diagonal_matrix (test/test-qrkit.cpp:63-131):
    numParams = 2*numVars, rows 7 per variable, values U(0.5,5) from default_random_engine in the reference's
    draw order; optional row shuffle (the reference uses std::random_shuffle; any shuffle exercises the same path)."""
    num_params = 2 * num_vars
    n_entries = sum(7 + (1 if (overlap and j < num_params - 2) else 0)
                    for i in range(num_params) for j in range(2 * i, min(2 * i + 2, num_params)))
    vals = orc.gen_uniform(seed, 0.5, 5.0, n_entries)
    rows, cols, k = [], [], 0
    for i in range(num_params):
        for j in range(2 * i, min(2 * i + 2, num_params)):
            for r in range(7):
                rows.append(7 * i + r); cols.append(j)
            if overlap and j < num_params - 2:
                rows.append(7 * i + 6); cols.append(j + 2)
    J = sp.csr_matrix((vals, (rows, cols)), shape=(7 * num_vars, num_params))
    if shuffle_seed is not None:
        p = np.random.default_rng(shuffle_seed).permutation(J.shape[0])
        J = J[p]
    J.sort_indices()
    return J


@pytest.mark.parametrize("overlap,shuffle", [(False, None), (True, None), (True, 3)])
def test_oracle_banded_invariants(overlap, shuffle):
    """test_banded_blocked (test/test-qrkit.cpp:208-258) on the oracle, SuggestedBlockCols = 8 as in the tests (:44):
    Q R = Pr J, Q^T Pr J = R, LS recovery."""
    J = banded_matrix(64, overlap, shuffle)
    res = orc.bb_factorize(J, suggested=8)
    n, m = J.shape
    inv = np.empty_like(res.row_perm); inv[res.row_perm] = np.arange(n)
    PJ = J[inv].toarray()
    R = res.R.toarray()
    assert rel_fro(orc.bb_apply_q(res, R, transpose=False), PJ) <= 1e-12       # Q R = Pr J   (:251)
    assert rel_fro(orc.bb_apply_q(res, PJ, transpose=True), R) <= 1e-12        # Q^T Pr J = R (:252)
    x = np.random.default_rng(0).uniform(-1, 1, m)
    y = orc.bb_apply_q(res, (J @ x)[inv], transpose=True)
    import scipy.linalg as sl
    assert rel_fro(sl.solve_triangular(R[:m, :m], y[:m]), x) <= 1e-10          # (:255), identity column permutation
    if overlap:
        assert len(res.blocks) == 21 and tuple(res.blocks[0]) == (0, 0, 21, 8)  # 7x4 strips merged to >= 8 columns


def test_product_structure_analysis_known_answers():
    """The PRODUCT's host-side analysis (qrk_bb_analyze_host) against the reference's known answers
    (test/test-utils.cpp:182-274) and against the oracle's block maps.  No GPU needed."""
    from qrkit_amd.banded import analyze_host
    from test_oracle_blockmap import block_diag_pattern, shuffled
    perm, blocks, has = analyze_host(shuffled(block_diag_pattern(256, False)), 2)
    i = np.arange(256)
    np.testing.assert_array_equal(blocks, np.stack([7 * i, 2 * i, np.full(256, 7), np.full(256, 2)], 1))
    assert has
    perm, blocks, has = analyze_host(shuffled(block_diag_pattern(256, True), 1), 2)
    assert len(blocks) == 255 and tuple(blocks[-1]) == (7 * 254, 2 * 254, 14, 4) and tuple(blocks[7]) == (49, 14, 7, 4)
    for overlap, shuf, sug in [(False, None, 2), (True, None, 8), (True, 5, 8), (True, 5, 3)]:
        J = banded_matrix(64, overlap, shuf)
        perm_o, blocks_o = orc.bb_analyze(J, sug)
        perm_p, blocks_p, _ = analyze_host(J, sug)
        np.testing.assert_array_equal(blocks_p, blocks_o)
        np.testing.assert_array_equal(perm_p, perm_o)
    # a pattern the reference cannot merge into portrait panels is rejected loudly
    from qrkit_amd import QrkError
    with pytest.raises(QrkError):
        analyze_host(sp.block_diag([np.ones((32, 32))] * 4, format="csr"), 2)


@pytest.mark.gpu
@pytest.mark.parametrize("num_vars,overlap,shuffle,suggested", [(256, False, None, 8), (256, True, None, 8), (256, True, 3, 8),
                                                                 (100, True, 7, 2), (64, True, None, 4)])
def test_hip_banded_matches_oracle(num_vars, overlap, shuffle, suggested):
    """test_banded_blocked's three inputs at the reference's size (256 variables, test-qrkit.cpp:369-385)."""
    import qrkit_amd
    J = banded_matrix(num_vars, overlap, shuffle)
    ref = orc.bb_factorize(J, suggested)
    qr = qrkit_amd.BandedBlockedSparseQR(suggestedBlockCols=suggested)
    qr.compute(J)
    np.testing.assert_array_equal(qr.blocks, ref.blocks)                          # bit-exact structure
    np.testing.assert_array_equal(qr.rowsPermutation(), ref.row_perm)
    R = qr.matrixR()
    np.testing.assert_array_equal(R.indptr, ref.R.indptr)
    np.testing.assert_array_equal(R.indices, ref.R.indices)
    # Sign of a row of R: where the leading entry of a reflector is exactly zero in exact arithmetic (the last
    # column of a merged panel: its only entry inside the panel is the overlap element), beta = -sign(x0)|x| takes
    # the sign of rounding noise, so a row of R (and the matching column of Q) may come out negated.  At most one
    # such row per panel; everything is compared after aligning those signs.
    n_rows, n_cols = J.shape
    dg, dgo = R.diagonal(), ref.R.diagonal()
    flip = np.where(np.sign(dg) != np.sign(dgo))[0]
    assert len(flip) <= len(ref.blocks), flip
    D = np.ones(n_rows); D[flip] = -1.0
    Rs = sp.diags(D) @ R
    Rs = sp.csc_matrix(Rs); Rs.sort_indices()
    assert rel_fro(Rs.toarray(), ref.R.toarray()) <= 1e-12
    # The merged panels of these inputs are numerically rank deficient in their last column (its only entry
    # inside the panel is the overlap element), so the last reflector of a panel is fixed by rounding noise and
    # Y/T are NOT comparable element by element between implementations (LAPACK disagrees with Eigen's algorithm
    # there as well).  What is comparable: R, the block descriptors, the range part of Q^T b, and the
    # reference's own invariants below.
    for k in (0, len(ref.yty) // 2, len(ref.yty) - 1):
        Y, T, row, nz = qr.blockYTY(k)
        Yo, To, rowo, nzo = ref.yty[k]
        assert (row, nz) == (rowo, nzo) and Y.shape == Yo.shape and T.shape == To.shape
        assert np.abs(np.triu(Y, 1)).max() == 0 and np.abs(np.diag(Y) - 1).max() == 0     # unit lower
        assert np.abs(np.tril(T, -1)).max() == 0                                          # upper triangular
        if not overlap:
            assert rel_fro(Y, Yo) <= 1e-12 and rel_fro(T, To) <= 1e-12
    n, m = J.shape
    inv = np.empty_like(ref.row_perm); inv[ref.row_perm] = np.arange(n)
    PJ = J[inv].toarray()
    Rd = R.toarray()
    assert rel_fro(qr.applyQ(Rd), PJ) <= 1e-12                                    # Q R = Pr J   (:251)
    assert rel_fro(qr.applyQt(PJ), Rd) <= 1e-12                                   # Q^T Pr J = R (:252)
    x = np.random.default_rng(0).uniform(-1, 1, m)
    b = np.random.default_rng(1).uniform(-1, 1, n)
    assert rel_fro(D[:m] * qr.applyQt(b)[:m], orc.bb_apply_q(ref, b, transpose=True)[:m]) <= 1e-11   # range part of Q^T b
    assert rel_fro(qr.solve((J @ x)[inv]), x) <= 1e-9                             # LS recovery  (:255)


def strip_matrix(num_strips, strip_rows, strip_cols, step_cols, seed=3):
    """Dense strips of strip_rows x strip_cols, strip i at rows strip_rows*i, columns step_cols*i (BASELINE configs[2]
    uses 256 x 192 strips stepping by 64 columns: SURVEY.md section 8d)."""
    rng = np.random.default_rng(seed)
    ncols = step_cols * num_strips
    rows, cols = [], []
    for i in range(num_strips):
        w = min(strip_cols, ncols - step_cols * i)
        r, c = np.meshgrid(np.arange(strip_rows * i, strip_rows * i + strip_rows), np.arange(step_cols * i, step_cols * i + w),
                           indexing="ij")
        rows.append(r.ravel()); cols.append(c.ravel())
    rows = np.concatenate(rows); cols = np.concatenate(cols)
    J = sp.csr_matrix((rng.uniform(0.5, 5.0, len(rows)), (rows, cols)), shape=(strip_rows * num_strips, ncols))
    J.sort_indices()
    return J


@pytest.mark.gpu
@pytest.mark.parametrize("strip_rows,env", [(256, None), (256, "QRK_BB_T_GLOBAL"), (256, "QRK_BB_CHAIN_V1"), (600, None)])
def test_hip_banded_baseline_shaped_panels(strip_rows, env, monkeypatch):
    """Panels of the BASELINE configs[2] shape (448 x 192 after merging: the 32-column blocks of the chain kernel), taller
    ones (16-column blocks), and the two fall-back paths, against the oracle and the reference's invariants."""
    import qrkit_amd
    if env:
        monkeypatch.setenv(env, "1")
    J = strip_matrix(6, strip_rows, 192, 64)
    ref = orc.bb_factorize(J, 2)
    qr = qrkit_amd.BandedBlockedSparseQR(suggestedBlockCols=2)
    qr.compute(J)
    np.testing.assert_array_equal(qr.blocks, ref.blocks)
    assert max(int(b[2]) for b in qr.blocks) >= strip_rows and max(int(b[3]) for b in qr.blocks) == 192
    R = qr.matrixR()
    np.testing.assert_array_equal(R.indptr, ref.R.indptr)
    np.testing.assert_array_equal(R.indices, ref.R.indices)
    assert rel_fro(R.toarray(), ref.R.toarray()) <= 1e-12
    for k in (0, len(ref.yty) - 1):
        Y, T, row, nz = qr.blockYTY(k)
        Yo, To, rowo, nzo = ref.yty[k]
        assert (row, nz) == (rowo, nzo)
        assert rel_fro(Y, Yo) <= 1e-12 and rel_fro(T, To) <= 1e-12
    n, m = J.shape
    Jd = J.toarray(); Rd = R.toarray()
    assert rel_fro(qr.applyQ(Rd), Jd) <= 1e-12                                    # Q R = J   (:251)
    assert rel_fro(qr.applyQt(Jd), Rd) <= 1e-12                                   # Q^T J = R (:252)
    x = np.random.default_rng(0).uniform(-1, 1, m)
    assert rel_fro(qr.solve(J @ x), x) <= 1e-10                                   # LS recovery (:255)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["reference", "strips"])
def test_hip_banded_solve_r_on_device(case):
    """qrk_bb_solve_r (back substitution of _solve_impl, :290-311) against a host triangular solve with the same R,
    several right-hand sides, both memory spaces."""
    import ctypes as C
    import scipy.sparse.linalg as spl
    import torch
    import qrkit_amd
    from qrkit_amd import _capi as capi
    J = banded_matrix(100, True, 7) if case == "reference" else strip_matrix(5, 256, 192, 64)
    qr = qrkit_amd.BandedBlockedSparseQR(suggestedBlockCols=2)
    qr.compute(J)
    rows, cols = J.shape
    R = qr.matrixR().tocsr()[:cols, :]
    rng = np.random.default_rng(5)
    Y = rng.uniform(-1, 1, (rows, 3))
    want = spl.spsolve_triangular(R, Y[:cols], lower=False)
    # device buffers
    v = torch.as_tensor(Y.T.copy(), device="cuda")                      # [nrhs, rows] = column-major, ld = rows
    capi.check(capi.lib().qrk_bb_solve_r(qr._plan, v.data_ptr(), rows, 3, capi.MEM_DEVICE), qr._ctx.handle)
    torch.cuda.synchronize()
    got = v.cpu().numpy()[:, :cols].T
    assert rel_fro(got, want) <= 1e-11
    np.testing.assert_array_equal(v.cpu().numpy()[:, cols:].T, Y[cols:])   # rows beyond cols untouched
    # host buffers
    h = np.asfortranarray(Y.copy())
    capi.check(capi.lib().qrk_bb_solve_r(qr._plan, h.ctypes.data_as(C.POINTER(C.c_double)), rows, 3, capi.MEM_HOST), qr._ctx.handle)
    np.testing.assert_array_equal(h[:cols], got)
    # the mirror's solve(): LS recovery with several right-hand sides (:255)
    X = rng.uniform(-1, 1, (cols, 2))
    inv = np.empty_like(qr.rowsPermutation()); inv[qr.rowsPermutation()] = np.arange(rows)
    assert rel_fro(qr.solve((J @ X)[inv]), X) <= 1e-9


@pytest.mark.gpu
def test_hip_banded_fixed_pattern_path():
    """The fixed-pattern analysis (BandedBlockedSparseQR.h:398-408: fixed-size block type and _BlockOverlap != Dynamic ->
    fromBlockBandedPattern, SparseQRUtils.h:274-302; qrk_bb_plan_create_fixed) on the reference's un-shuffled overlapping matrix:
    the block map is the oracle's and the reference's known answer (test-utils.cpp:228-241), and the factorisation equals the
    generic-analysis one bit for bit (same block map, identity row permutation)."""
    import qrkit_amd
    from qrkit_amd.banded import blocks_from_pattern
    J = banded_matrix(256, True, None)
    fixed = qrkit_amd.BandedBlockedSparseQR(suggestedBlockCols=8, fixedPattern=(7, 4, 2))
    generic = qrkit_amd.BandedBlockedSparseQR(suggestedBlockCols=8)
    fixed.compute(J)
    generic.compute(J)
    np.testing.assert_array_equal(fixed.blocks, orc.from_block_banded_pattern(J.shape[0], J.shape[1], 7, 4, 2, 8))
    np.testing.assert_array_equal(fixed.blocks, blocks_from_pattern(J.shape[0], J.shape[1], 7, 4, 2, 8))
    np.testing.assert_array_equal(fixed.blocks, generic.blocks)
    assert not fixed.hasPermutation
    Rf, Rg = fixed.matrixR(), generic.matrixR()
    np.testing.assert_array_equal(Rf.indptr, Rg.indptr)
    np.testing.assert_array_equal(Rf.indices, Rg.indices)
    np.testing.assert_array_equal(Rf.data, Rg.data)
