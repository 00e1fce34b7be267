"""Time of solve() of the block-diagonal solver (Q^T b, per-tile back substitution, permutation) for uniform batches, device vectors.
Usage (GPU box): python tools/solve_probe.py"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import qrkit_amd as qa
ctx = qa.Context(0)
for (r, c, B) in ((32, 32, 10000), (32, 32, 100000), (7, 2, 1000000), (8, 6, 20000), (8, 6, 1000000), (64, 64, 20000), (256, 256, 1000)):
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    tiles = torch.rand(B * r * c, device="cuda", dtype=torch.float64) * 2 - 1
    mat = qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles)
    qr = qa.BlockDiagonalSparseQR(context=ctx)
    qr.compute(mat)
    b = torch.rand(B * r, device="cuda", dtype=torch.float64)
    x = qr.solve(b); torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        x = qr.solve(b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    byts = B * (8 * r * r + 4 * c * (c + 1) + 8 * r + 8 * c + 4 * c)
    print(f"{r:3d}x{c:<3d} B={B:8d}  solve {dt*1e6:9.1f} us  {byts/dt/1e9:8.1f} GB/s ({byts/dt/8e12*100:4.1f} % of 8 TB/s: Q, R, b read once, x written)", flush=True)
