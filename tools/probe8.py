import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import qrkit_amd as qa
ctx = qa.Context(0)
for (r, c, b) in ((7, 2, 2000000), (6, 6, 1000000), (8, 6, 1000000), (4, 4, 2000000), (8, 8, 1000000)):
    rows, cols = np.full(b, r, np.int32), np.full(b, c, np.int32)
    tiles = torch.rand(b * r * c, device="cuda", dtype=torch.float64) * 2 - 1
    mat = qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles)
    qr = qa.BlockDiagonalSparseQR(context=ctx)
    qr.analyzePattern(mat); qr.factorize(mat); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): qr.factorize(mat)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"{r}x{c} {dt*1e6:9.1f} us  {b/dt/1e6:8.1f} M tiles/s", flush=True)
