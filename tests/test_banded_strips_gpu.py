"""GPU: the strips form of the banded factorisation (qrk_bbs_*: BandedBlockedSparseQR::factorize, src/QRKit/BandedBlockedSparseQR.h:
463-508, on a block-banded matrix handed over as dense strips; stage A = every strip triangularised on all CUs, stage B = a chain
that merges two triangles per step).  R is unique up to the signs of its rows for a fixed column order, so the
checks are: R against LAPACK's QR of the assembled matrix after aligning row signs, against the oracle's restatement of the reference's
chain the same way, Q^T J = R, Q Q^T b = b, and the least-squares solution."""
import os

import numpy as np
import pytest

from helpers import rel_fro


def assemble(strips, N, ms, n, s):
    J = np.zeros((N * ms, (N - 1) * s + n))
    for i in range(N):
        J[i * ms:(i + 1) * ms, i * s:i * s + n] = strips[i]
    return J


def make(N, ms, n, s, seed):
    rng = np.random.default_rng(seed)
    return rng.uniform(-1.0, 1.0, (N, ms, n)) + 0.25 * np.sign(rng.uniform(-1, 1, (N, ms, n)))


def to_device(strips):
    import torch
    # strip i column-major
    return torch.from_numpy(np.ascontiguousarray(strips.transpose(0, 2, 1)).reshape(-1)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("N,ms,n,s", [(1, 64, 48, 16), (2, 64, 48, 16), (5, 96, 64, 32), (6, 256, 192, 64), (9, 40, 32, 16),
                                      (4, 128, 96, 96), (3, 256, 256, 64)])
def test_strips_r_matches_lapack_and_products(N, ms, n, s):
    import torch
    import qrkit_amd
    from qrkit_amd.banded import BandedStripsQR
    strips = make(N, ms, n, s, seed=N * 7 + n)
    J = assemble(strips, N, ms, n, s)
    rows, cols = J.shape
    qr = BandedStripsQR(N, ms, n, s, context=qrkit_amd.Context(0))
    qr.factorize(to_device(strips))
    R = qr.matrixR_dense()
    assert np.abs(np.tril(R, -1)).max() == 0.0
    # band structure: row r of R has nothing right of its window
    for i in range(N - 1):
        assert np.abs(R[i * s:(i + 1) * s, i * s + n:]).max(initial=0.0) == 0.0
    Rl = np.linalg.qr(J, mode="r")
    sg = np.sign(np.diag(R)) * np.sign(np.diag(Rl))
    assert np.all(sg != 0)
    row_err = np.linalg.norm(R * sg[:, None] - Rl, axis=1) / np.linalg.norm(Rl, axis=1)
    assert row_err.max() <= 1e-11, row_err.max()
    # Q^T J = [R; 0] in the documented layout (the rows of R first)
    Jd = torch.from_numpy(np.asfortranarray(J).T.copy()).cuda().t()
    QtJ = qr.applyQ(Jd, transpose=True).cpu().numpy()
    assert np.linalg.norm(QtJ[:cols] - R) <= 1e-12 * np.linalg.norm(J) * np.sqrt(cols)
    assert np.linalg.norm(QtJ[cols:]) <= 1e-12 * np.linalg.norm(J) * np.sqrt(cols)
    # Q Q^T b = b, |Q^T b| = |b|
    b = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, rows)).cuda()
    qtb = qr.applyQ(b, transpose=True)
    assert abs(float(qtb.norm()) - float(b.norm())) <= 1e-12 * float(b.norm())
    assert rel_fro(qr.applyQ(qtb, transpose=False).cpu().numpy(), b.cpu().numpy()) <= 1e-12
    # least squares through the implicit Q
    x = np.random.default_rng(2).uniform(-1, 1, cols)
    xs = qr.solve(torch.from_numpy(J @ x).cuda()).cpu().numpy()
    assert rel_fro(xs, x) <= 1e-9
    bb = np.random.default_rng(3).uniform(-1, 1, (rows, 3))
    xls = qr.solve(torch.from_numpy(np.asfortranarray(bb).T.copy()).cuda().t()).cpu().numpy()
    ref = np.linalg.lstsq(J, bb, rcond=None)[0]
    assert rel_fro(xls, ref) <= 1e-9


@pytest.mark.gpu
def test_strips_r_matches_the_reference_chain_up_to_row_signs():
    """The same matrix through the reference's own elimination order (the oracle's restatement of BandedBlockedSparseQR, and the CSR
    entry qrk_bb_* that follows it): R agrees after aligning the sign of each row."""
    import scipy.sparse as sp
    import torch
    import qrkit_amd
    from qrkit_amd.banded import BandedStripsQR
    from oracle import oracle as orc
    N, ms, n, s = 6, 64, 48, 16
    strips = make(N, ms, n, s, seed=3)
    J = assemble(strips, N, ms, n, s)
    qr = BandedStripsQR(N, ms, n, s, context=qrkit_amd.Context(0))
    qr.factorize(to_device(strips))
    R = qr.matrixR_dense()
    res = orc.bb_factorize(sp.csr_matrix(J), suggested=s)
    assert not np.any(res.row_perm != np.arange(J.shape[0]))
    Ro = res.R.toarray()[:J.shape[1]]
    sg = np.sign(np.diag(R)) * np.sign(np.diag(Ro))
    row_err = np.linalg.norm(R * sg[:, None] - Ro, axis=1) / np.linalg.norm(Ro, axis=1)
    assert row_err.max() <= 1e-11, row_err.max()


@pytest.mark.gpu
def test_strips_configs2_full_size():
    """BASELINE configs[2]: 50 000 strips of 256 x 192, column step 64 (12.8 M x 3.2 M, 2.46e9 stored entries), at FULL size by
    default (about 50 s and 45 GB of HBM; QRK_BIG_STRIPS=<n> runs fewer).  Properties: Q^T J = R on sampled strips (columns of J are
    sparse: one strip wide), |Q^T b| = |b|, LS recovery."""
    import time
    import torch
    import qrkit_amd
    from qrkit_amd.banded import BandedStripsQR
    N, ms, n, s = int(os.environ.get("QRK_BIG_STRIPS", "50000")), 256, 192, 64
    g = torch.Generator(device="cuda").manual_seed(5)
    strips = torch.rand(N * ms * n, device="cuda", dtype=torch.float64, generator=g) * 2 - 1
    qr = BandedStripsQR(N, ms, n, s, context=qrkit_amd.Context(0))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    qr.factorize(strips)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"configs[2] strips form: {N} strips in {dt:.3f} s = {dt / N * 1e3:.4f} ms per strip")
    rows, cols = qr.rows(), qr.cols()
    x = torch.rand(cols, device="cuda", dtype=torch.float64, generator=g) * 2 - 1
    # b = J x, strip by strip
    S3 = strips.view(N, n, ms)
    b = torch.empty(rows, device="cuda", dtype=torch.float64)
    for i0 in range(0, N, 2000):
        i1 = min(N, i0 + 2000)
        idx = (torch.arange(i0, i1, device="cuda") * s)[:, None] + torch.arange(n, device="cuda")[None, :]
        b.view(N, ms)[i0:i1] = torch.einsum("inm,in->im", S3[i0:i1], x[idx])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    xs = qr.solve(b)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    xs2 = qr.solve(b)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"configs[2] solve: first call {t1 - t0:.3f} s (builds the maps of banded_maps.hip), then {t2 - t1:.3f} s = {(t2 - t1) / N * 1e6:.2f} us per strip")
    assert torch.equal(xs, xs2)
    err = float((xs - x).norm() / x.norm())
    print(f"configs[2] LS recovery: {err:.2e}")
    assert err <= 1e-8
    qtb = qr.applyQ(b, transpose=True)
    assert abs(float(qtb.norm()) - float(b.norm())) <= 1e-11 * float(b.norm())
    assert float(qtb[cols:].norm()) <= 1e-9 * float(b.norm())       # b is in the range of J
    # Q^T (column c of J) = column c of R, on sampled columns
    for c in (0, 63, 64, 1000, cols // 2, cols - 193, cols - 1):
        e = torch.zeros(rows, device="cuda", dtype=torch.float64)
        for i in range(max(0, (c - n) // s), min(N, c // s + 1)):
            if i * s <= c < i * s + n:
                e[i * ms:(i + 1) * ms] = S3[i, c - i * s]
        q = qr.applyQ(e, transpose=True)
        col = torch.zeros(cols, device="cuda", dtype=torch.float64)
        for i in range(max(0, (c - n) // s), min(N, c // s + 1)):
            blk = qr.rRows(i)
            if i * s <= c < i * s + n:
                col[i * s:i * s + blk.shape[0]] = blk[:, c - i * s]
        assert float((q[:cols] - col).norm()) <= 1e-11 * float(e.norm()), c
        assert float(q[cols:].norm()) <= 1e-11 * float(e.norm()), c


@pytest.mark.gpu
def test_given_up_pipeline_falls_back_to_one_workgroup(capfd):
    """A wait of the pipelined chain that runs out (QRK_BBS_PIPE_SPINS=0: the first unsatisfied wait) raises the chain's abort word;
    qrk_bbs_factorize reads it at its synchronisation point and runs stage B again on one workgroup: same R as the single-workgroup
    chain, never a silently unfinished factor (round-4 advisor finding on banded.hip's bounded wait)."""
    import qrkit_amd
    from qrkit_amd.banded import BandedStripsQR
    N, ms, n, s = 24, 256, 192, 64
    strips = make(N, ms, n, s, seed=77)
    dev = to_device(strips)
    old = {k: os.environ.get(k) for k in ("QRK_BBS_PIPE", "QRK_BBS_PIPE_SPINS", "QRK_BBS_PIPE_VERBOSE")}
    try:
        os.environ["QRK_BBS_PIPE"] = "1"
        qr1 = BandedStripsQR(N, ms, n, s, context=qrkit_amd.Context(0))
        qr1.factorize(dev)
        R1 = qr1.matrixR_dense()
        os.environ["QRK_BBS_PIPE"] = "3"
        os.environ["QRK_BBS_PIPE_SPINS"] = "0"
        os.environ["QRK_BBS_PIPE_VERBOSE"] = "1"
        qr3 = BandedStripsQR(N, ms, n, s, context=qrkit_amd.Context(0))
        qr3.factorize(dev)
        R3 = qr3.matrixR_dense()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    err = capfd.readouterr().err
    assert "ran again on one workgroup" in err, "the shortened wait did not trigger the fall-back"
    assert np.array_equal(R1, R3), "the fall-back differs from the single-workgroup chain"
    Rl = np.linalg.qr(assemble(strips, N, ms, n, s), mode="r")
    sg = np.sign(np.diag(R3)) * np.sign(np.diag(Rl))
    assert (np.linalg.norm(R3 * sg[:, None] - Rl, axis=1) / np.linalg.norm(Rl, axis=1)).max() <= 1e-11


@pytest.mark.gpu
@pytest.mark.parametrize("N,ms,n,s", [(2, 64, 48, 16), (3, 64, 48, 16), (9, 40, 32, 16), (12, 96, 64, 32), (24, 256, 192, 64), (7, 256, 256, 64),
                                      (6, 256, 192, 96), (5, 256, 240, 16), (40, 128, 128, 64), (70, 64, 48, 16), (131, 96, 64, 32),
                                      (100, 256, 192, 64), (66, 256, 240, 16),
                                      # an overlap SHORTER than the column step (0 < lo < s): the carry maps serve Q^T b / Q x, the
                                      # triangular solve must stay on the one-workgroup chain (its map form needs lo >= s)
                                      (8, 128, 96, 64), (9, 64, 64, 48), (7, 96, 80, 64), (70, 128, 96, 64)])
def test_carry_maps_agree_with_the_one_workgroup_chains(N, ms, n, s):
    """banded_maps.hip: Q^T b, Q x and the triangular solve through one small matrix per strip (M_i = the linear map of the carry,
    G_i = D_i^-1 U_i) against the chains that walk the strips on one workgroup (QRK_BBS_MAPS=0): the same operators associated
    differently, so agreement to rounding -- one and several right-hand sides, carries of 16 ... 224 entries (both forms of the chain
    kernel), column steps above 64 (the triangular solve stays on the old kernel there), 64 strips and more (the chains run in two
    levels: groups of K strips at once, then the group boundaries through the products of the groups' maps; QRK_BBS_MAPS_K), and a
    second factorisation on the same plan (the maps are rebuilt)."""
    import torch
    import qrkit_amd
    from qrkit_amd.banded import BandedStripsQR
    old = os.environ.get("QRK_BBS_MAPS")

    def both(fn):
        out = []
        try:
            for sw in ("0", "1"):
                os.environ["QRK_BBS_MAPS"] = sw
                # (the scratch and the result of a call come out of torch's caching allocator: make them NaN, not the other form's
                #  answer -- an entry one form fails to write must not be masked by what the other left in the same block)
                torch.cuda.synchronize(); torch.cuda.empty_cache()
                junk = torch.full((1 << 22,), float("nan"), dtype=torch.float64, device="cuda")
                del junk
                out.append(fn().cpu().numpy())
        finally:
            if old is None:
                os.environ.pop("QRK_BBS_MAPS", None)
            else:
                os.environ["QRK_BBS_MAPS"] = old
        return out

    qr = BandedStripsQR(N, ms, n, s, context=qrkit_amd.Context(0))
    for rep in range(2):
        strips = make(N, ms, n, s, seed=100 * rep + N + n)
        rows, cols = N * ms, (N - 1) * s + n
        J = assemble(strips, N, ms, n, s) if rows * cols <= 4e6 else None
        qr.factorize(to_device(strips))
        rng = np.random.default_rng(rep)
        for nrhs in (1, 5):
            b = torch.from_numpy(np.asfortranarray(rng.uniform(-1, 1, (rows, nrhs))).T.copy()).cuda().t()
            q0, q1 = both(lambda: qr.applyQ(b, transpose=True))
            assert rel_fro(q1, q0) <= 1e-13, (rep, nrhs)
            y = torch.from_numpy(np.asfortranarray(q0).T.copy()).cuda().t()
            z0, z1 = both(lambda: qr.applyQ(y, transpose=False))
            assert rel_fro(z1, z0) <= 1e-13, (rep, nrhs)
            assert rel_fro(z1, b.cpu().numpy()) <= 1e-12
            x0, x1 = both(lambda: qr.solve(b))
            assert rel_fro(x1, x0) <= 1e-10, (rep, nrhs)
            if J is not None:
                ref = np.linalg.lstsq(J, b.cpu().numpy(), rcond=None)[0]
                assert rel_fro(x1, ref) <= 1e-9


@pytest.mark.gpu
def test_maps_that_cannot_be_allocated_leave_the_old_chains_in_charge():
    """bbs_maps() gives the per-strip maps up for good when their memory is not there (13 GB at the full configs[2] size) and the plan keeps
    answering through the one-workgroup chains: QRK_BBS_MAPS_FAIL forces that branch on the first product; the answers stay right, with
    the switch removed as well (maps_off is a property of the plan), and a fresh plan goes through the maps again."""
    import torch
    import qrkit_amd
    from qrkit_amd.banded import BandedStripsQR
    N, ms, n, s = 70, 64, 48, 16
    strips = make(N, ms, n, s, seed=5)
    J = assemble(strips, N, ms, n, s)
    b = np.random.default_rng(0).uniform(-1, 1, J.shape[0])
    ref = np.linalg.lstsq(J, b, rcond=None)[0]
    old = os.environ.get("QRK_BBS_MAPS_FAIL")
    qr = BandedStripsQR(N, ms, n, s, context=qrkit_amd.Context(0))
    qr.factorize(to_device(strips))
    try:
        os.environ["QRK_BBS_MAPS_FAIL"] = "1"
        x1 = qr.solve(torch.from_numpy(b).cuda()).cpu().numpy()
    finally:
        if old is None:
            os.environ.pop("QRK_BBS_MAPS_FAIL", None)
        else:
            os.environ["QRK_BBS_MAPS_FAIL"] = old
    x2 = qr.solve(torch.from_numpy(b).cuda()).cpu().numpy()
    assert rel_fro(x1, ref) <= 1e-9 and rel_fro(x2, ref) <= 1e-9
    assert np.array_equal(x1, x2)                       # the same kernels both times: the plan gave the maps up
    qr2 = BandedStripsQR(N, ms, n, s, context=qrkit_amd.Context(0))
    qr2.factorize(to_device(strips))
    x3 = qr2.solve(torch.from_numpy(b).cuda()).cpu().numpy()
    assert rel_fro(x3, ref) <= 1e-9 and rel_fro(x3, x1) <= 1e-10
    assert not np.array_equal(x3, x1)                   # (the maps associate the products differently: equal to rounding, not bitwise)
