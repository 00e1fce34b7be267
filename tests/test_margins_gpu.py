"""GPU: walk the decision margins.  The fast kernels take a pivot decision only when it is clear of their own rounding (squared norms,
FMA chains: qrk_device.h `decide`, bdqr_pair.hip "Decisions and the exact path") and hand the tile to the exact path otherwise.  The
tie battery (test_ties_gpu.py) has exact ties and clear-cut data; here the two leading candidates of a step are placed a GRADED
distance apart -- k ulp in norm for k = 0, 1, 2, 4 .. 2^30 -- at step 0 and after s earlier pivots whose reflectors have already
downdated both candidates.  Asserted for every kernel family: the permutation is the oracle's for EVERY k (a margin that is too small
is the one way left to return a wrong permutation silently), and a gap inside the kernel's own error bound is flagged (the tile then
is bitwise the oracle's).  The smallest gap that was decided on the fast path is reported per family (printed, and kept within the
band the margin was designed for).

Construction (one tile): columns 0..s-1 are large (norms 8 .. 8 (1 + s/10): clear pivots, supported on the top rows); columns s and
s + 1 are the candidates: the same magnitudes entry by entry -- equal top parts, so the s reflectors downdate both by the same
amounts, and sign-flipped bottom parts -- with the second one scaled by (1 + k eps): their norms differ by k ulp, before and after
the downdates; the remaining columns are small."""
import numpy as np
import pytest

from helpers import oracle_factorize, per_tile_rel, tile_sizes

pytestmark = pytest.mark.gpu

KS = [0, 1, 2, 4, 16, 256, 2 ** 10, 2 ** 12, 2 ** 13, 2 ** 14, 2 ** 15, 2 ** 16, 2 ** 18, 2 ** 20, 2 ** 24, 2 ** 30]
PER_K = 12


def margin_tiles(r, c, s, seed):
    """len(KS) * PER_K tiles r x c (column-major, packed), tile (ik * PER_K + t) has its candidates k = KS[ik] ulp apart."""
    rng = np.random.default_rng(seed)
    eps = np.finfo(np.float64).eps
    top = max(s, r // 2)                              # the large columns live on rows < top (s of them: full rank there)
    assert s + 2 <= c and top < r and s <= top
    tiles = np.zeros((len(KS) * PER_K, c, r))
    for ik, k in enumerate(KS):
        for t in range(PER_K):
            a = tiles[ik * PER_K + t]
            a[:] = rng.uniform(-0.1, 0.1, (c, r))
            for j in range(s):
                a[j] = 0.0
                a[j, :top] = rng.uniform(-1, 1, top)
                a[j] *= 8.0 * (1.0 + 0.1 * (s - j)) / np.linalg.norm(a[j])
            u = rng.uniform(0.5, 1.0, r) * rng.choice([-1.0, 1.0], r)
            u *= 2.0 / np.linalg.norm(u)
            v = u.copy()
            v[top:] *= rng.choice([-1.0, 1.0], r - top)
            if np.array_equal(v, u):
                v[r - 1] = -v[r - 1]
            a[s] = u
            a[s + 1] = v * (1.0 + k * eps)
            # where the two candidates sit among the columns must not matter
            perm = rng.permutation(c)
            a[:] = a[perm]
    return np.ascontiguousarray(tiles).reshape(-1)


FAMILIES = [
    # name, rows, cols, steps s
    ("K1 pair kernel 32x32", 32, 32, [0, 8, 24]),
    ("K1 ragged 24x20", 24, 20, [0, 8]),
    ("K5 small 8x6", 8, 6, [0, 2]),
    ("K5 small 16x16", 16, 16, [0, 4, 8]),
    ("thin 9x2", 9, 2, [0]),
    ("K2 one wave per tile 48x48", 48, 48, [0, 8, 24]),
    ("K2 one wave per tile 64x64", 64, 64, [0, 16, 40]),
    ("K2 LDS-resident 80x48", 80, 48, [0, 8, 24]),
    ("K2 on-chip 200x200", 200, 200, [0, 8, 24]),
]


@pytest.mark.parametrize("name,r,c,steps", FAMILIES, ids=[f[0].replace(" ", "_") for f in FAMILIES])
def test_margin_walk(name, r, c, steps, capsys):
    import qrkit_amd
    ctx = qrkit_amd.Context(0)
    report = []
    for s in steps:
        per_k = PER_K if r * c <= 4096 else 3
        B = len(KS) * PER_K
        tiles = margin_tiles(r, c, s, seed=1000 + 7 * r + s)
        if per_k != PER_K:                                 # large tiles: fewer tiles per gap
            keep = np.concatenate([np.arange(ik * PER_K, ik * PER_K + per_k) for ik in range(len(KS))])
            tiles = tiles.reshape(B, c * r)[keep].reshape(-1)
            B = len(keep)
        rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
        mat = qrkit_amd.SparseBlockDiagonal.fromTiles(rows, cols, tiles)
        qr = qrkit_amd.BlockDiagonalSparseQR(mat, context=ctx)
        _, ref = oracle_factorize(rows, cols, tiles)
        got = qr.colsPermutation()
        got = got.cpu().numpy() if hasattr(got, "cpu") else np.asarray(got)
        bad = np.nonzero(np.any(got.reshape(B, c) != ref.perm.reshape(B, c), axis=1))[0]
        assert len(bad) == 0, f"{name}, step {s}: permutation differs from the oracle on tiles {bad[:8]} (gaps {[KS[b // per_k] for b in bad[:8]]} ulp)"
        sq, sr, sc = tile_sizes(rows, cols)
        Q = qr.qValues().cpu().numpy()
        assert per_tile_rel(Q, ref.Q_vals, sq) <= 1e-12
        flagged = np.array([np.array_equal(Q[i * r * r:(i + 1) * r * r], ref.Q_vals[i * r * r:(i + 1) * r * r]) for i in range(B)])
        frac = flagged.reshape(len(KS), per_k).mean(axis=1)
        # inside the error bound of the kernel's own norms (k <= 16 ulp is far inside 2^14 eps): every tile flagged
        for ik, k in enumerate(KS):
            if k <= 16:
                assert frac[ik] == 1.0, f"{name}, step {s}: a gap of {k} ulp was decided on the fast path"
        unflagged = [KS[ik] for ik in range(len(KS)) if frac[ik] < 1.0]
        smallest = min(unflagged) if unflagged else None
        # ... and far outside it (2^24 ulp = 4e-9 relative) the candidates themselves are no reason to flag: at step 0 nothing is
        # flagged (after many downdates the small remaining columns can meet Eigen's norm-recompute test near its threshold, a
        # decision of its own: reported, not asserted)
        if s == 0:
            assert frac[KS.index(2 ** 30)] == 0.0 and frac[KS.index(2 ** 24)] == 0.0, f"{name}, step {s}: well-separated candidates were flagged"
        report.append(f"{name:28s} step {s:2d}: smallest gap decided on the fast path = {smallest} ulp; flagged fraction by gap: " +
                      " ".join(f"{k}:{f:.2f}" for k, f in zip(KS, frac)))
    with capsys.disabled():
        for line in report:
            print("\n[margins] " + line, end="")


@pytest.mark.parametrize("path,two_stage", [("cols", "0"), (None, "1")], ids=["dense_direct", "dense_two_stage"])
def test_margin_walk_dense_solver(path, two_stage, monkeypatch, capsys):
    """The same walk for the dense right-block solver (qrk_dense_*, BlockAngularSparseQR.h:361-369): one 640 x 48 block per gap, the
    candidates meet at step 0 and at step 8.  Direct column-parallel kernel (Eigen's format: a flagged block is bitwise the oracle's)
    and the two-stage form (R up to row signs: flagged = the exact path's Eigen format, recognisable by its reflectors)."""
    import torch
    import qrkit_amd
    from qrkit_amd.angular import DenseColPivQR
    from oracle import oracle as orc
    monkeypatch.setenv("QRK_DENSE_TWO_STAGE", two_stage)
    if path:
        monkeypatch.setenv("QRK_DENSE_PATH", path)
    r, c = 640, 48
    ctx = qrkit_amd.Context(0)
    lines = []
    for s in (0, 8):
        flagged = []
        tiles = margin_tiles(r, c, s, seed=77 + s).reshape(len(KS) * PER_K, c, r)
        for ik, k in enumerate(KS):
            A = np.ascontiguousarray(tiles[ik * PER_K].T)             # r x c
            qr = DenseColPivQR(ctx, 0)
            At = torch.from_numpy(np.asfortranarray(A).T.copy()).cuda().t()
            qr.compute(At)
            ref, hc, perm, _ = orc.colpiv_qr(A)
            np.testing.assert_array_equal(qr.colsPermutation().cpu().numpy(), perm, err_msg=f"gap {k} ulp at step {s}")
            got = At.cpu().numpy()
            flagged.append(bool(np.array_equal(np.tril(got, -1), np.tril(ref, -1))))      # Eigen's reflectors, bit for bit
            Rg, Rr = np.triu(got[:c]), np.triu(ref[:c])
            sg = np.sign(np.diag(Rg)) * np.sign(np.diag(Rr))
            assert np.linalg.norm(Rg * sg[:, None] - Rr) <= 1e-11 * np.linalg.norm(Rr)
        for ik, k in enumerate(KS):
            if k <= 16:
                assert flagged[ik], f"dense solver, step {s}: a gap of {k} ulp was decided on the fast path"
        un = [k for k, f in zip(KS, flagged) if not f]
        lines.append(f"dense solver ({'two-stage' if two_stage == '1' else 'direct'}) step {s}: smallest gap decided on the fast path = {min(un) if un else None} ulp")
    with capsys.disabled():
        for line in lines:
            print("\n[margins] " + line, end="")
