"""Time of the block-local auxiliary entry points of the block-diagonal solver on device vectors: applyQt (y = Q^T b), applyQ, solveR (the
per-tile back substitution alone), pattern() (CSR of Q / CSC of R).  Usage (GPU box): python tools/aux_probe.py"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import qrkit_amd as qa
ctx = qa.Context(0)


def tm(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for (r, c, B) in ((32, 32, 10000), (7, 2, 1000000), (8, 6, 20000), (8, 6, 1000000), (16, 16, 400000), (64, 64, 20000)):
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    tiles = torch.rand(B * r * c, device="cuda", dtype=torch.float64) * 2 - 1
    qr = qa.BlockDiagonalSparseQR(context=ctx)
    qr.compute(qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles))
    b = torch.rand(B * r, device="cuda", dtype=torch.float64)
    y = torch.rand(B * c, device="cuda", dtype=torch.float64)
    qb = B * (8 * r * r + 16 * r)
    rb = B * (4 * c * (c + 1) + 16 * c)
    t1 = tm(lambda: qr.applyQt(b)); t2 = tm(lambda: qr.applyQ(b)); t3 = tm(lambda: qr.solveR(y))
    print(f"{r:3d}x{c:<3d} B={B:8d}  applyQt {t1*1e6:8.1f} us ({qb/t1/8e12*100:4.1f} %)  applyQ {t2*1e6:8.1f} us ({qb/t2/8e12*100:4.1f} %)  "
          f"solveR {t3*1e6:8.1f} us ({rb/t3/8e12*100:4.1f} %)   (% of 8 TB/s at Q or R read once + the vectors)", flush=True)
