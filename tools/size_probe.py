"""Throughput of uniform square tiles of the sizes given on the command line (default 96 128), B tiles each.
Usage (GPU box): [QRK_COL_THREADS=256] python tools/size_probe.py 96 128"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import qrkit_amd as qa
ctx = qa.Context(0)
for s in [int(a) for a in sys.argv[1:]] or [96, 128]:
    b = max(500, min(20000, 16_000_000 // (s * s)))
    rows = np.full(b, s, np.int32)
    tiles = torch.rand(b * s * s, device="cuda", dtype=torch.float64) * 2 - 1
    mat = qa.SparseBlockDiagonal.fromTiles(rows, rows, tiles)
    qr = qa.BlockDiagonalSparseQR(blockSolver=0, qFormat=0, context=ctx)
    qr.analyzePattern(mat); qr.factorize(mat); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): qr.factorize(mat)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{s}x{s} B={b}: {dt*1e3:.3f} ms  {b/dt:.0f} tiles/s", flush=True)
