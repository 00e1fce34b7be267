"""Kernel-rate probe of the small-tile kernel at batch sizes where host overhead does not matter.
Usage (GPU box): python tools/small_probe_big.py"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import qrkit_amd as qa

ctx = qa.Context(0)
SHAPES = ((7, 2, 2000000), (9, 2, 2000000), (6, 6, 1000000), (8, 6, 1000000), (4, 4, 2000000), (8, 8, 1000000),
          (12, 12, 400000), (16, 16, 400000))
if len(sys.argv) > 1:      # e.g. "9x2,12x2,16x1"
    SHAPES = tuple((int(a.split("x")[0]), int(a.split("x")[1]), 1000000) for a in sys.argv[1].split(","))
for (r, c, b) in SHAPES:
    rows, cols = np.full(b, r, np.int32), np.full(b, c, np.int32)
    tiles = torch.rand(b * r * c, device="cuda", dtype=torch.float64) * 2 - 1
    mat = qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles)
    qr = qa.BlockDiagonalSparseQR(context=ctx)
    qr.analyzePattern(mat)
    qr.factorize(mat); torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        qr.factorize(mat)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    byts = b * (8 * r * c + 8 * r * r + 4 * c * (c + 1) + 4 * c)
    print(f"{r:2d}x{c:<2d} B={b:8d}  {dt*1e6:9.1f} us  {b/dt/1e6:9.1f} M tiles/s  {byts/dt/1e9:8.1f} GB/s ({byts/dt/8e12*100:4.1f} % of 8 TB/s)", flush=True)
