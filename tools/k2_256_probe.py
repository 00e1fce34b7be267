"""K2 probe for the counters: B (default 1024) tiles of 256 x 256 through BlockDiagonalSparseQR (ColPiv, Full Q), a few factorisations.
Usage (GPU box): python tools/k2_256_probe.py [B] [reps]   -- under rocprofv3 --pmc for profiles/r03_k2_pmc.txt"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import qrkit_amd as qa
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
s = 256
ctx = qa.Context(0)
rows = np.full(b, s, np.int32)
tiles = torch.rand(b * s * s, device="cuda", dtype=torch.float64) * 2 - 1
mat = qa.SparseBlockDiagonal.fromTiles(rows, rows, tiles)
qr = qa.BlockDiagonalSparseQR(blockSolver=0, qFormat=0, context=ctx)
qr.analyzePattern(mat)
qr.factorize(mat); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    qr.factorize(mat)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"{s}x{s} B={b} {dt*1e3:.3f} ms/launch {b/dt:.0f} tiles/s {8*s**3/3*b/dt/1e12:.2f} TFLOP/s (8 n^3/3 per tile, SURVEY.md 8(d): R and the explicit Q)", flush=True)
