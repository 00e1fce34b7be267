import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import qrkit_amd as qa
from oracle import oracle as orc
ctx = qa.Context(0)
rng = np.random.default_rng(0)
for n in (33, 48, 64, 96):
    B = 200
    tiles = rng.uniform(0.5, 5.0, B * n * n)
    rows = np.full(B, n, np.int32); cols = np.full(B, n, np.int32)
    mat = qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles)
    qr = qa.BlockDiagonalSparseQR(mat, context=ctx)
    ref = orc.BDProblem(rows, cols, tiles).factorize()
    got = qr.rValues().cpu().numpy()
    per = n * (n + 1) // 2
    same = sum(np.array_equal(got[i*per:(i+1)*per], ref.R_vals[i*per:(i+1)*per]) for i in range(B))
    print(f"{n}x{n}: tiles bitwise equal to the oracle (= went through the exact path): {same} of {B}", flush=True)
