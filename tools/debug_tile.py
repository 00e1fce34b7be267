import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import qrkit_amd
from oracle import oracle as orc
B, r, c = 4, 32, 32
tiles = orc.gen_uniform(1, 0.5, 5.0, B*r*c)
rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
for solver in (1, 0):
    mat = qrkit_amd.SparseBlockDiagonal.fromTiles(rows, cols, tiles)
    qr = qrkit_amd.BlockDiagonalSparseQR(blockSolver=solver); qr.compute(mat)
    ref = orc.BDProblem(rows, cols, tiles, block_solver=solver).factorize()
    P = qr.colsPermutation()
    print("solver", solver, "perm equal:", np.array_equal(P, ref.perm))
    if not np.array_equal(P, ref.perm):
        print(P[:32]); print(ref.perm[:32])
    R = qr.rValues().cpu().numpy().reshape(B, -1); Rr = ref.R_vals.reshape(B, -1)
    li = np.tril_indices(c)
    Rm = np.zeros((c, c)); Rm[li[1], li[0]] = R[0]; Rf = np.zeros((c, c)); Rf[li[1], li[0]] = Rr[0]
    err = np.abs(Rm - Rf)
    print("R err by row (tile 0):", np.array2string(err.max(axis=1), precision=1))
    Q = qr.qValues().cpu().numpy().reshape(B, r, r)[0]; Qf = ref.Q_vals.reshape(B, r, r)[0]
    print("Q err max:", np.abs(Q - Qf).max())
    E = np.abs(Q - Qf)
    print("Q err by row:", np.array2string(E.max(axis=1), precision=1))
    print("Q err by col:", np.array2string(E.max(axis=0), precision=1))
    print("Q orth:", np.abs(Q.T @ Q - np.eye(r)).max(), " Q[0,:4]", Q[0,:4], "ref", Qf[0,:4])
