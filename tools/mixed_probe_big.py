"""Timing probe for the tiles of K2 that work in global memory (65...256) and the mixed batch: tools/mixed_probe.py without
the small sizes.  Usage (GPU box): python tools/mixed_probe_big.py [B]"""
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
    print(f"{label:40s} B={len(rows):6d}  {dt*1e3:9.3f} ms  {len(rows)/dt:12.0f} tiles/s", flush=True)


B = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
n = rng.integers(8, 257, B).astype(np.int32)
timeit(n, n, "mixed square 8..256")
for s, b in ((80, 1000), (96, 1000), (128, 1000), (160, 500), (192, 500), (224, 500), (256, 500), (256, 2000)):
    timeit(np.full(b, s, np.int32), np.full(b, s, np.int32), f"uniform {s}x{s}")
