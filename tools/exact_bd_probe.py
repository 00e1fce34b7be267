"""Throughput of the exact-arithmetic path for block-diagonal batches: +-1 tiles (every pivot decision is a tie, so EVERY tile is
redone in Eigen's operation order) against generic tiles of the same shape.  Usage (GPU box): python tools/exact_bd_probe.py"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import qrkit_amd as qa
ctx = qa.Context(0)
for (r, c, B) in ((32, 32, 10000), (8, 6, 20000), (16, 16, 20000), (64, 64, 2000), (128, 128, 500)):
    for kind in ("generic", "pm1"):
        if kind == "pm1":
            t = (torch.randint(0, 2, (B * r * c,), device="cuda").double() * 2 - 1)
        else:
            t = torch.rand(B * r * c, device="cuda", dtype=torch.float64) * 2 - 1
        rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
        mat = qa.SparseBlockDiagonal.fromTiles(rows, cols, t)
        qr = qa.BlockDiagonalSparseQR(blockSolver=0, qFormat=0, context=ctx)
        qr.analyzePattern(mat)
        qr.factorize(mat); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            qr.factorize(mat)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"{r:3d}x{c:<3d} B={B:6d} {kind:8s} {dt * 1e3:10.3f} ms  {B / dt:14.0f} tiles/s", flush=True)
