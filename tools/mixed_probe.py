"""Timing probe for mixed-size batches (BASELINE configs[4] shape, reduced count) and single sizes.
Usage (GPU box): python tools/mixed_probe.py [B]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import qrkit_amd as qa

ctx = qa.Context(0)
rng = np.random.default_rng(1)


def timeit(rows, cols, label, reps=3):
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
    flops = (2 * r * c * c - 2 * c ** 3 / 3 + 4 * (r * r * c - r * c * c + c ** 3 / 3)).sum()
    print(f"{label:40s} B={len(rows):6d}  {dt*1e3:9.3f} ms  {len(rows)/dt:12.0f} tiles/s  {byts/dt/1e9:8.1f} GB/s  {flops/dt/1e9:8.1f} GFLOP/s", flush=True)


B = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
n = rng.integers(8, 257, B).astype(np.int32)
timeit(n, n, "mixed square 8..256")
for s, b in ((32, 10000), (33, 2000), (33, 20000), (40, 20000), (48, 2000), (48, 20000), (56, 20000), (64, 2000), (64, 10000), (64, 40000), (96, 1000), (128, 1000), (192, 500), (256, 500)):
    timeit(np.full(b, s, np.int32), np.full(b, s, np.int32), f"uniform {s}x{s}")
