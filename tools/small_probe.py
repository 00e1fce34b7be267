"""Timing probe for small uniform tiles (the left parts of the compositions: 6x6, 8x6, 7x2) and ragged <= 32.
Usage (GPU box): python tools/small_probe.py"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import qrkit_amd as qa

ctx = qa.Context(0)
rng = np.random.default_rng(1)


def timeit(rows, cols, label, reps=20):
    n_in = int((rows.astype(np.int64) * cols).sum())
    tiles = torch.rand(n_in, device="cuda", dtype=torch.float64) * 2 - 1
    mat = qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles)
    qr = qa.BlockDiagonalSparseQR(blockSolver=0, qFormat=0, context=ctx)
    qr.analyzePattern(mat)
    qr.factorize(mat); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        qr.factorize(mat)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    r = rows.astype(np.float64); c = cols.astype(np.float64)
    byts = (8 * r * c + 8 * r * r + 4 * c * (c + 1) + 4 * c).sum()
    print(f"{label:34s} B={len(rows):7d}  {dt*1e6:9.1f} us  {len(rows)/dt/1e6:8.2f} M tiles/s  {byts/dt/1e9:8.1f} GB/s", flush=True)


for (r, c, b) in ((6, 6, 20000), (8, 6, 20000), (7, 2, 100000), (16, 16, 20000), (24, 24, 20000), (32, 20, 20000), (32, 32, 20000)):
    timeit(np.full(b, r, np.int32), np.full(b, c, np.int32), f"uniform {r}x{c}")
n = rng.integers(2, 33, 20000).astype(np.int32)
timeit(n, n, "ragged square 2..32")
