import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import oracle_factorize, seeded_tiles
import qrkit_amd
for (B, r, c) in ((8, 8, 8), (8, 8, 6), (3, 8, 6), (16, 6, 6)):
    tiles = seeded_tiles(5, -1.0, 1.0, B * r * c)
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    _, ref = oracle_factorize(rows, cols, tiles)
    qr = qrkit_amd.BlockDiagonalSparseQR(blockSolver=0, qFormat=0, context=qrkit_amd.Context(0))
    qr.compute(qrkit_amd.SparseBlockDiagonal.fromTiles(rows, cols, tiles))
    P = qr.colsPermutation().reshape(B, c); Pr = ref.perm.reshape(B, c)
    Q = qr.qValues().cpu().numpy().reshape(B, -1); Qr = ref.Q_vals.reshape(B, -1)
    R = qr.rValues().cpu().numpy().reshape(B, -1); Rr = ref.R_vals.reshape(B, -1)
    print(f"== {B} x ({r} x {c})")
    for i in range(B):
        print(i, "perm ok" if np.array_equal(P[i], Pr[i]) else f"perm {P[i] - i * c} vs {Pr[i] - i * c}",
              "R err %.2e" % (np.linalg.norm(R[i] - Rr[i]) / np.linalg.norm(Rr[i])), "Q err %.2e" % (np.linalg.norm(Q[i] - Qr[i]) / np.linalg.norm(Qr[i])),
              "R bitwise" if np.array_equal(R[i], Rr[i]) else "")
