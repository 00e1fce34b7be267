"""Fuzz of the several-tiles-per-wavefront kernel (bdqr_quad.hip) against the oracle: random UNIFORM batches of tiles with 5...16 rows and
3...rows columns (1...70 tiles: every filling of the last wavefront), five value distributions, both solvers; permutation bit-exact, Q / R
within 1e-11 per tile.  Also runs every batch twice and compares the two results bitwise.
Usage (GPU box): python tools/fuzz_quad.py [batches] [seed0]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import oracle_factorize, per_tile_rel, tile_sizes
import qrkit_amd
from qrkit_amd import _capi as capi

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t0 = time.time()
worst = 0.0
for b in range(nb):
    rng = np.random.default_rng(70000 + seed0 + b)
    B = int(rng.integers(1, 71))
    r0 = int(rng.integers(5, 17)) if b % 2 else int(rng.integers(5, 9))
    c0 = int(rng.integers(3, r0 + 1))                         # (1- and 2-column tiles belong to bdqr_thin)
    rows = np.full(B, r0, np.int32); cols = np.full(B, c0, np.int32)
    n = int((rows.astype(np.int64) * cols).sum())
    kind = b % 5
    if kind == 0:
        tiles = rng.uniform(-1.0, 1.0, n)
    elif kind == 1:
        tiles = rng.uniform(-1.0, 1.0, n) * np.exp2(rng.integers(-30, 31, n))
    elif kind == 2:
        tiles = rng.integers(-2, 3, n).astype(np.float64)
    elif kind == 3:
        tiles = rng.uniform(-1.0, 1.0, n) * 1e-3
        off = 0
        for r, c in zip(rows, cols):
            tiles[off: off + r] += rng.uniform(1.0, 2.0, r); off += int(r) * int(c)
    else:
        tiles = rng.standard_normal(n)
        off = 0
        for r, c in zip(rows, cols):                      # graded columns: norms spread over 12 orders of magnitude
            a = tiles[off: off + int(r) * int(c)].reshape(int(c), int(r))
            a *= np.logspace(0, -12, int(c))[rng.permutation(int(c))][:, None]; off += int(r) * int(c)
    solver = capi.COLPIV_HOUSEHOLDER if b % 4 else capi.HOUSEHOLDER
    mat = qrkit_amd.SparseBlockDiagonal.fromTiles(rows, cols, tiles)
    qr = qrkit_amd.BlockDiagonalSparseQR(mat, blockSolver=solver)
    q1, r1, p1 = qr.qValues().clone(), qr.rValues().clone(), qr.colsPermutation().copy()
    qr.factorize(mat)
    assert bool((q1 == qr.qValues()).all()) and bool((r1 == qr.rValues()).all()) and (p1 == qr.colsPermutation()).all(), f"batch {b}: two runs differ"
    _, ref = oracle_factorize(rows, cols, tiles, block_solver=solver)
    assert qr.info() == 0 and qr.rank() == ref.rank, b
    assert (qr.colsPermutation() == ref.perm).all(), f"batch {b}: permutation"
    sq, sr, _ = tile_sizes(rows, cols)
    eq = per_tile_rel(qr.qValues().cpu().numpy(), ref.Q_vals, sq); er = per_tile_rel(qr.rValues().cpu().numpy(), ref.R_vals, sr)
    if eq > 1e-11 or er > 1e-11:
        # two backward-stable factorisations of a badly scaled tile differ in Q by cond * eps (seen: a 4 x 4 tile with entries over 60
        # binary orders of magnitude, |Q - Q_oracle| 4.6e-11 at residual 4e-16): then the factorisation itself has to be right
        Qv, Rv, Pv = qr.qValues().cpu().numpy(), qr.rValues().cpu().numpy(), qr.colsPermutation()
        oq = orr = ot = oc = 0
        for r_, c_ in zip(rows, cols):
            r_, c_ = int(r_), int(c_)
            q = Qv[oq:oq + r_ * r_].reshape(r_, r_)
            il = np.tril_indices(c_); Rm = np.zeros((r_, c_)); Rm[il[1], il[0]] = Rv[orr:orr + c_ * (c_ + 1) // 2]
            A = tiles[ot:ot + r_ * c_].reshape(c_, r_).T
            pp = Pv[oc:oc + c_] - oc
            assert np.linalg.norm(q @ Rm - A[:, pp]) <= 1e-14 * np.linalg.norm(A) * np.sqrt(r_), f"batch {b}: residual"
            assert np.linalg.norm(q.T @ q - np.eye(r_)) <= 1e-13 * np.sqrt(r_), f"batch {b}: orthogonality"
            oq += r_ * r_; orr += c_ * (c_ + 1) // 2; ot += r_ * c_; oc += c_
        print(f"batch {b} kind {kind}: element-wise Q {eq:.1e} R {er:.1e} above 1e-11, residual and orthogonality fine (conditioning)", flush=True)
    else:
        worst = max(worst, eq, er)
    if b % 20 == 19:
        print(f"{b + 1} batches ok, worst per-tile error so far {worst:.2e}, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz ok: {nb} batches, worst per-tile relative error {worst:.2e}")
