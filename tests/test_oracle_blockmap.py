"""Integer known answers of the reference's structure analysis (test/test-utils.cpp:182-274) on the oracle,
plus the simulated cases of SURVEY.md Appendix C."""
import numpy as np
import scipy.sparse as sp

from oracle import oracle as orc


def block_diag_pattern(num_vars, overlap):
    """Pattern of generate_block_diagonal_matrix / generate_overlapping_block_diagonal_matrix
    (test/test-utils.cpp:39-103): numParams = 2*numVars columns, 7 rows per variable."""
    num_params = 2 * num_vars
    rows, cols = [], []
    for i in range(num_params):
        for j in range(2 * i, min(2 * i + 2, num_params)):
            for r in range(7):
                rows.append(7 * i + r); cols.append(j)
            if overlap and j < num_params - 2:
                rows.append(7 * i + 6); cols.append(j + 2)
    m = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(7 * num_vars, num_params))
    m.sort_indices()
    return m


def shuffled(m, seed=0):
    p = np.random.default_rng(seed).permutation(m.shape[0])
    out = m[p]
    out.sort_indices()
    return out


def abap_then_blocks(m, suggested=2):
    has, perm = orc.as_banded_as_possible(m.shape[0], m.shape[1], m.indptr, m.indices)
    if has:
        inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
        m = m[inv]       # (P*M).row(perm[i]) = M.row(i)
        m.sort_indices()
    return orc.block_info_from_csr(m.shape[0], m.shape[1], m.indptr, m.indices, suggested)


def test_blockdiag_permuted():                       # test-utils.cpp:182-209
    b = abap_then_blocks(shuffled(block_diag_pattern(256, False)))
    assert len(b) == 256
    i = np.arange(256)
    np.testing.assert_array_equal(b, np.stack([7 * i, 2 * i, np.full(256, 7), np.full(256, 2)], 1))


def test_overlapping_permuted():                     # test-utils.cpp:211-252
    b = abap_then_blocks(shuffled(block_diag_pattern(256, True), 1))
    assert len(b) == 255
    i = np.arange(255)
    want = np.stack([7 * i, 2 * i, np.full(255, 7), np.full(255, 4)], 1)
    want[-1, 2] = 14
    np.testing.assert_array_equal(b, want)


def test_blockdiag_vertperm_diag():                  # test-utils.cpp:145-180, 254-274
    J = block_diag_pattern(256, False).tocsc()
    n_res, n_par = J.shape
    # rowpermADiagLambda: lambda row of column c goes right below the last nonzero of column c
    perm = np.zeros(n_res + n_par, dtype=np.int64)
    curr = 0
    for c in range(n_par):
        col_rows = J.indices[J.indptr[c]:J.indptr[c + 1]]
        last = col_rows[-1] if len(col_rows) else 0
        while curr <= last + c:
            perm[curr - c] = curr; curr += 1
        perm[n_res + c] = curr; curr += 1
    stacked = sp.vstack([J, sp.identity(n_par)]).tocsr()
    inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
    m = stacked[inv]
    m.sort_indices()
    b = orc.block_info_from_csr(m.shape[0], m.shape[1], m.indptr, m.indices, 2)
    i = np.arange(256)
    np.testing.assert_array_equal(b, np.stack([9 * i, 2 * i, np.full(256, 9), np.full(256, 2)], 1))


def test_from_block_diagonal_pattern():              # SparseQRUtils.h:255-272
    b = orc.from_block_diagonal_pattern(7 * 256, 2 * 256, 7, 2)
    i = np.arange(256)
    np.testing.assert_array_equal(b, np.stack([7 * i, 2 * i, np.full(256, 7), np.full(256, 2)], 1))
    assert len(orc.from_block_diagonal_pattern(100, 33, 32, 32)) == 1     # numBlocks = cols / blockCols


def test_from_block_banded_pattern_cases():          # SparseQRUtils.h:274-302 + mergeBlocks :308-385
    b = orc.from_block_banded_pattern(7 * 256, 2 * 256, 7, 4, 2, 2)
    assert len(b) == 255 and tuple(b[-1]) == (7 * 254, 2 * 254, 14, 4) and tuple(b[3]) == (21, 6, 7, 4)
    b8 = orc.from_block_banded_pattern(7 * 256, 2 * 256, 7, 4, 2, 8)     # the tests' SuggestedBlockCols = 8
    assert len(b8) == 85 and tuple(b8[0]) == (0, 0, 21, 8) and tuple(b8[-1]) == (21 * 84, 6 * 84, 28, 8)
    # square / landscape strips make the reference call back() on an empty vector (UB): reported as None
    assert orc.from_block_banded_pattern(3200, 3200, 64, 192, 128, 2) is None
    assert orc.from_block_banded_pattern(3200, 3200, 32, 32, 0, 2) is None
    assert orc.from_block_banded_pattern(3300, 3200, 33, 32, 0, 2) is not None
