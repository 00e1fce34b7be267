"""GPU: the three generations of the uniform 32 x 32 kernel -- bdqr_quad32.hip (QRK_K1_FORM=quad32: four tiles per wave, one per DPP row,
two columns per lane), bdqr_pair4.hip (QRK_K1_FORM=pair4: two tiles per wave, four waves per SIMD, pivot column published to LDS, Q by
backward accumulation; the plan picks one of these two by launch size) and bdqr_pair.hip (QRK_PAIR_V2=0: LDS image of A, Q^T carried along) -- against
the oracle on the same inputs: permutation bit-exact, Q / R / tau within 1e-12 per tile; odd tile counts (the last wave has one tile),
both block solvers, tie families (the flagged tiles are redone inside the kernel by the exact routine), and at BASELINE's size the
size-independent properties.  QRK_PAIR_V2 / QRK_K1_FORM are read when the plan is created."""
import numpy as np
import pytest

from helpers import oracle_factorize, seeded_tiles
from test_ties_gpu import KINDS, check, tie_tiles

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def qa():
    import qrkit_amd
    return qrkit_amd


@pytest.fixture(scope="module")
def ctx(qa):
    return qa.Context(0)


def select_generation(monkeypatch, gen):
    if gen == "gen1":
        monkeypatch.setenv("QRK_PAIR_V2", "0")
        monkeypatch.delenv("QRK_K1_FORM", raising=False)
    else:
        monkeypatch.delenv("QRK_PAIR_V2", raising=False)
        monkeypatch.setenv("QRK_K1_FORM", "quad32" if gen == "gen3" else "pair4")


@pytest.fixture(params=["gen3", "gen2", "gen1"])
def generation(request, monkeypatch):
    select_generation(monkeypatch, request.param)
    return request.param


def kernel_of(qa, ctx, B, solver):
    import ctypes as C
    from qrkit_amd import _capi as capi
    lay = capi.BDLayout()
    lay.num_blocks, lay.block_rows, lay.block_cols = B, 32, 32
    lay.rows = lay.cols = None
    lay.mat_rows = lay.mat_cols = B * 32
    plan = C.c_void_p()
    capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), 0, solver, C.byref(plan)))
    capi.lib().qrk_bd_kernel_name.restype = C.c_char_p
    capi.lib().qrk_bd_kernel_name.argtypes = [C.c_void_p, C.c_int]
    name = capi.lib().qrk_bd_kernel_name(plan, 0).decode()
    capi.lib().qrk_bd_plan_destroy(plan)
    return name


def test_generation_switch_selects_the_kernel(qa, ctx, generation):
    name = kernel_of(qa, ctx, 10, 0)
    assert ("bdqr_pair4_kernel" in name) == (generation == "gen2"), name
    assert ("bdqr_quad32_kernel" in name) == (generation == "gen3"), name


@pytest.mark.parametrize("solver", [0, 1])
@pytest.mark.parametrize("B", [1, 2, 3, 4, 5, 7, 129, 1000])
def test_generic_batches_against_the_oracle(qa, ctx, generation, B, solver):
    tiles = seeded_tiles(100 + B, 0.5, 5.0, B * 1024)
    rows = cols = np.full(B, 32, np.int32)
    mat = qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles)
    qr = qa.BlockDiagonalSparseQR(context=ctx, blockSolver=solver)
    qr.compute(mat)
    _, ref = oracle_factorize(rows, cols, tiles, block_solver=solver)
    check(qr, ref, rows, cols)


@pytest.mark.parametrize("kind", KINDS)
def test_tie_families(qa, ctx, generation, kind):
    B = 301                                           # (odd: the last wave of the launch has one tile)
    tiles = tie_tiles(kind, B, 32, 32, seed=4000 + KINDS.index(kind))
    rows = cols = np.full(B, 32, np.int32)
    qr = qa.BlockDiagonalSparseQR(qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles), context=ctx)
    _, ref = oracle_factorize(rows, cols, tiles)
    check(qr, ref, rows, cols)


def test_degenerate_tiles(qa, ctx, generation):
    """Zero tiles, zero and duplicate columns, a zero first column without pivoting, tiny and huge scales."""
    rng = np.random.default_rng(11)
    B = 64
    a = rng.uniform(-1, 1, size=(B, 32, 32))          # [tile][column][row]
    a[0] = 0.0
    a[1, 5] = 0.0; a[1, 9] = a[1, 3]
    a[2, :, 10:] = 0.0                                # rank 10
    a[3] *= 1e-150; a[4] *= 1e150
    a[5, 0] = 0.0                                     # zero first column: H = I for the unpivoted solver
    a[6, :, 0] = 0.0                                  # zero first row: x0 = 0 at step 0
    a[7] = np.eye(32)
    a[8] = np.triu(a[8])
    tiles = np.ascontiguousarray(a).reshape(-1)
    rows = cols = np.full(B, 32, np.int32)
    for solver in (0, 1):
        qr = qa.BlockDiagonalSparseQR(context=ctx, blockSolver=solver)
        qr.compute(qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles))
        _, ref = oracle_factorize(rows, cols, tiles, block_solver=solver)
        check(qr, ref, rows, cols)


def test_generations_agree_and_are_deterministic(qa, ctx, monkeypatch):
    """Same permutation from the three kernels, values within 1e-12 of each other, each kernel bitwise reproducible run to run; the two
    two-phase kernels (gen2 without the own-norm step -- one round here -- and gen3) compute the same products in the same order:
    bitwise the same Q and R."""
    B = 4097
    tiles = seeded_tiles(5, -1.0, 1.0, B * 1024)
    rows = cols = np.full(B, 32, np.int32)
    mat = qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles)
    out = {}
    for gen in ("gen3", "gen2", "gen1"):
        select_generation(monkeypatch, gen)
        runs = []
        for _ in range(2):
            qr = qa.BlockDiagonalSparseQR(context=ctx)
            qr.compute(mat)
            runs.append((qr.colsPermutation().copy(), qr.qValues().cpu().numpy().copy(), qr.rValues().cpu().numpy().copy()))
        for x, y in zip(runs[0], runs[1]):
            np.testing.assert_array_equal(x, y)
        out[gen] = runs[0]
    np.testing.assert_array_equal(out["gen2"][0], out["gen1"][0])
    np.testing.assert_allclose(out["gen2"][1], out["gen1"][1], rtol=0, atol=1e-12)
    scale = np.abs(out["gen1"][2]).max()
    np.testing.assert_allclose(out["gen2"][2], out["gen1"][2], rtol=0, atol=1e-12 * scale)
    for x, y in zip(out["gen3"], out["gen2"]):
        np.testing.assert_array_equal(x, y)


def test_baseline_size_properties(qa, ctx, generation):
    """configs[1]: 10 000 tiles of 32 x 32 -- Q^T Q = I, Q R = A P, every tile's permutation a permutation of its own columns."""
    import torch
    B = 10000
    g = torch.Generator(device="cuda").manual_seed(3)
    t = torch.rand(B * 1024, device="cuda", dtype=torch.float64, generator=g) * 4.5 + 0.5
    rows = cols = np.full(B, 32, np.int32)
    qr = qa.BlockDiagonalSparseQR(context=ctx)
    qr.compute(qa.SparseBlockDiagonal.fromTiles(rows, cols, t))
    Q = qr.qValues().view(B, 32, 32)
    A = t.view(B, 32, 32).transpose(1, 2)
    R = torch.zeros(B, 32, 32, device="cuda", dtype=torch.float64)
    iu = torch.triu_indices(32, 32, device="cuda")
    order = torch.argsort(iu[1] * 32 + iu[0])
    R[:, iu[0][order], iu[1][order]] = qr.rValues().view(B, 528)
    perm = torch.as_tensor(qr.colsPermutation(), device="cuda").view(B, 32).long()
    base = (torch.arange(B, device="cuda") * 32)[:, None]
    P = perm - base
    assert bool(((P >= 0) & (P < 32)).all()) and bool((torch.sort(P, dim=1).values == torch.arange(32, device="cuda")).all())
    AP = torch.gather(A, 2, P[:, None, :].expand(B, 32, 32))
    eye = torch.eye(32, device="cuda", dtype=torch.float64)
    assert (Q.transpose(1, 2) @ Q - eye).abs().max().item() < 1e-13
    assert ((Q @ R - AP).abs().amax(dim=(1, 2)) / A.abs().amax(dim=(1, 2))).max().item() < 1e-13
    d = torch.diagonal(R, dim1=1, dim2=2).abs()
    assert bool((d[:, :-1] >= d[:, 1:] * (1 - 1e-9)).all())          # pivoted: |R_kk| non-increasing


@pytest.mark.parametrize("gen", ["gen3", "gen2"])
@pytest.mark.parametrize("wgs,B", [(16, 3001), (7, 1200)])
def test_many_rounds_per_workgroup(qa, ctx, monkeypatch, wgs, B, gen):
    """Few workgroups (QRK_PAIR_WGS / QRK_Q32_WGS): a workgroup runs more than the 32 rounds whose flags one word remembers -- several
    chunks of rounds, each followed by the exact redo of its flagged tiles; tie tiles in every chunk."""
    select_generation(monkeypatch, gen)
    monkeypatch.setenv("QRK_PAIR_WGS", str(wgs))
    monkeypatch.setenv("QRK_Q32_WGS", str(wgs))
    rng = np.random.default_rng(B)
    a = rng.uniform(-1, 1, size=(B, 32, 32))
    ties = rng.choice(B, size=B // 7, replace=False)
    a[ties] = rng.choice([-1.0, 1.0], size=(len(ties), 32, 32))
    tiles = np.ascontiguousarray(a).reshape(-1)
    rows = cols = np.full(B, 32, np.int32)
    qr = qa.BlockDiagonalSparseQR(qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles), context=ctx)
    _, ref = oracle_factorize(rows, cols, tiles)
    check(qr, ref, rows, cols)


def test_eight_byte_aligned_arrays(qa, ctx, generation):
    """The C ABI takes any double-aligned arrays: tiles, Q and R that start 8 bytes off a 16-byte boundary (16-byte loads and stores of
    the kernels at 8-byte alignment) give the same result as aligned ones."""
    import ctypes as C
    import torch
    from qrkit_amd import _capi as capi
    B = 257
    lay = capi.BDLayout()
    lay.num_blocks, lay.block_rows, lay.block_cols = B, 32, 32
    lay.rows = lay.cols = None
    lay.mat_rows = lay.mat_cols = B * 32
    plan = C.c_void_p()
    capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), capi.BLOCK_DIAGONAL_Q, capi.COLPIV_HOUSEHOLDER, C.byref(plan)))
    g = torch.Generator(device="cuda").manual_seed(17)
    out = []
    for shift in (0, 1):
        tbuf = torch.zeros(B * 1024 + 2, device="cuda", dtype=torch.float64)
        g.manual_seed(17)
        tbuf[shift:shift + B * 1024] = torch.rand(B * 1024, device="cuda", dtype=torch.float64, generator=g) * 4.5 + 0.5
        qbuf = torch.zeros(B * 1024 + 2, device="cuda", dtype=torch.float64)
        rbuf = torch.zeros(B * 528 + 2, device="cuda", dtype=torch.float64)
        pm = torch.zeros(B * 32, device="cuda", dtype=torch.int32)
        capi.check(capi.lib().qrk_bd_factorize(plan, tbuf.data_ptr() + 8 * shift, qbuf.data_ptr() + 8 * shift, rbuf.data_ptr() + 8 * shift,
                                               pm.data_ptr(), None, capi.MEM_DEVICE))
        torch.cuda.synchronize()
        assert qbuf[:shift].abs().sum().item() == 0.0 and qbuf[shift + B * 1024:].abs().sum().item() == 0.0     # nothing outside
        assert rbuf[:shift].abs().sum().item() == 0.0 and rbuf[shift + B * 528:].abs().sum().item() == 0.0
        out.append((qbuf[shift:shift + B * 1024].clone(), rbuf[shift:shift + B * 528].clone(), pm.clone()))
    capi.lib().qrk_bd_plan_destroy(plan)
    for x, y in zip(out[0], out[1]):
        assert torch.equal(x, y)
