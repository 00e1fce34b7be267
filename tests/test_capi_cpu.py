"""CPU tests of the product's host side: the C-ABI library loads and exports every declared symbol,
fails loudly without a GPU, and the host-side integer logic matches the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "qrkit_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(qrk_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from qrkit_amd import _capi
    lib = _capi.lib()
    names = header_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/qrkit_amd.h but not exported"
    assert sorted(_capi.EXPORTS) == names
    assert lib.qrk_version() == 1


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from qrkit_amd import _capi
    lib = _capi.lib()
    assert lib.qrk_device_count() == 0
    h = C.c_void_p()
    st = lib.qrk_create(C.byref(h), 0, None)
    assert st == _capi.STATUS_NO_DEVICE and not h
    assert b"no CPU fallback" in lib.qrk_last_error(None)
    import qrkit_amd
    with pytest.raises(RuntimeError):
        qrkit_amd.Context(0)


def test_product_does_not_import_oracle():
    """The product path must never route through oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "qrkit_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                assert "oracle" not in open(os.path.join(dirpath, f), errors="ignore").read().lower(), f
    for dirpath, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            assert "oracle" not in open(os.path.join(dirpath, f), errors="ignore").read().lower(), f


def test_from_block_diagonal_pattern_matches_oracle():
    import scipy.sparse as sp
    from oracle import oracle as orc
    from qrkit_amd.solvers import SparseBlockDiagonal
    nv = 40
    tiles = orc.gen_reference_7x2(nv)
    J = sp.block_diag([tiles[i * 14:(i + 1) * 14].reshape(2, 7).T for i in range(nv)], format="csc")
    mat = SparseBlockDiagonal().fromBlockDiagonalPattern(J, 7, 2)
    bm = orc.from_block_diagonal_pattern(J.shape[0], J.shape[1], 7, 2)
    assert mat.size() == len(bm) == nv and mat.rows() == 7 * nv and mat.cols() == 2 * nv
    np.testing.assert_array_equal(mat.block_rows, bm[:, 2])
    np.testing.assert_array_equal(mat.block_cols, bm[:, 3])
    got = mat.tiles if mat.tiles is not None else mat.tiles_dev.cpu().numpy()      # (cut on the device when there is one)
    np.testing.assert_array_equal(got, tiles)
    np.testing.assert_array_equal(mat[3], tiles[42:56].reshape(2, 7).T)


def test_shard_ranges_are_contiguous_and_balanced():
    from qrkit_amd.sharding import shard_offsets, shard_ranges
    rows = np.full(10000, 32); cols = np.full(10000, 32)
    for w in (1, 2, 4, 8):
        rg = shard_ranges(rows, cols, w)
        assert rg[0][0] == 0 and rg[-1][1] == 10000
        assert all(a[1] == b[0] for a, b in zip(rg, rg[1:]))
        assert max(e - s for s, e in rg) - min(e - s for s, e in rg) <= 1
    rng = np.random.default_rng(0)
    n = rng.integers(8, 257, 5000)
    rg = shard_ranges(n, n, 8)
    cost = np.array([(n[s:e].astype(float) ** 3).sum() for s, e in rg])
    assert cost.max() / cost.mean() < 1.05
    br, bc, qo, ro = shard_offsets(n, n, *rg[3])
    assert br == n[:rg[3][0]].sum() and qo == (n[:rg[3][0]].astype(np.int64) ** 2).sum()


def test_tiles_from_sparse_rejects_null_arguments():
    from qrkit_amd import _capi
    lib = _capi.lib()
    assert lib.qrk_bd_tiles_from_sparse(None, 0, None, None, None, 0, None, _capi.MEM_HOST) == _capi.STATUS_INVALID_ARGUMENT


def test_blocks_from_pattern_matches_oracle_and_known_answers():
    """qrk_bb_blocks_from_pattern (host logic of the fixed-pattern banded path: BlockBandedMatrixInfo::fromBlockBandedPattern +
    mergeBlocks, src/QRKit/SparseQRUtils.h:274-385) against the oracle and the reference's known answer
    (test/test-utils.cpp:228-241: 7x4 blocks, overlap 2 -> 255 blocks (7i, 2i, 7, 4), the last one 14x4)."""
    from oracle import oracle as orc
    from qrkit_amd.banded import blocks_from_pattern
    b = blocks_from_pattern(7 * 256, 2 * 256, 7, 4, 2, 2)
    assert len(b) == 255
    assert all(tuple(b[i]) == (7 * i, 2 * i, 7, 4) for i in range(254)) and tuple(b[254]) == (7 * 254, 2 * 254, 14, 4)
    for args in [(7 * 256, 2 * 256, 7, 4, 2, 2), (7 * 256, 2 * 256, 7, 4, 2, 8), (3300, 3200, 33, 32, 0, 2), (90, 40, 9, 4, 2, 3)]:
        want = orc.from_block_banded_pattern(*args)
        assert want is not None
        np.testing.assert_array_equal(blocks_from_pattern(*args), want)


def test_c_shard_ranges_match_the_python_mirror():
    """qrk_shard_ranges (the C ABI's partition of the diagonal blocks over the ranks: host integer logic, no GPU) against
    qrkit_amd.sharding.shard_ranges / shard_offsets on uniform, mixed and degenerate layouts."""
    from qrkit_amd import _capi
    from qrkit_amd.sharding import shard_offsets, shard_ranges
    lib = _capi.lib()
    lib.qrk_shard_ranges.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.POINTER(_capi.Shard)]
    rng = np.random.default_rng(3)
    cases = [(np.full(10000, 32, np.int32), np.full(10000, 32, np.int32))]
    for B in (1, 2, 7, 100, 5000):
        c = rng.integers(1, 257, B).astype(np.int32)
        cases.append(((c + rng.integers(0, 9, B)).astype(np.int32), c))
    cases.append((np.zeros(0, np.int32), np.zeros(0, np.int32)))
    for rows, cols in cases:
        for world in (1, 2, 3, 8, 13):
            sh = (_capi.Shard * (world + 1))()
            st = lib.qrk_shard_ranges(len(rows), 0, 0, rows.ctypes.data_as(C.POINTER(C.c_int32)), cols.ctypes.data_as(C.POINTER(C.c_int32)), world, sh)
            assert st == _capi.STATUS_OK
            want = shard_ranges(rows, cols, world)
            assert [(s.first_block, s.first_block + s.num_blocks) for s in sh[:world]] == [tuple(map(int, w)) for w in want]
            for g, (a, b) in enumerate(want):
                br, bc, qo, ro = shard_offsets(rows, cols, a, b)
                assert (sh[g].base_row, sh[g].base_col, sh[g].q_off, sh[g].r_off) == (br, bc, qo, ro)
                assert sh[g].tiles_off == int((rows[:a].astype(np.int64) * cols[:a]).sum())
            assert sh[world].first_block == len(rows) and sh[world].num_blocks == 0
            assert sh[world].r_off == int((cols.astype(np.int64) * (cols + 1) // 2).sum())
    # the uniform form (rows = cols = NULL)
    sh = (_capi.Shard * 9)()
    assert lib.qrk_shard_ranges(10000, 32, 32, None, None, 8, sh) == _capi.STATUS_OK
    assert [s.num_blocks for s in sh[:8]] == [1250] * 8 and sh[8].tiles_off == 10000 * 1024
    assert lib.qrk_shard_ranges(10, 0, 0, None, None, 2, sh) == _capi.STATUS_INVALID_ARGUMENT
    assert lib.qrk_gather_r(None, None, 0, 1, 0, sh, None, None, None, None) == _capi.STATUS_INVALID_ARGUMENT
    assert lib.qrk_gather_x(None, None, 0, 1, 0, sh, None, 1, None) == _capi.STATUS_INVALID_ARGUMENT
