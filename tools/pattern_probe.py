"""Time of analyzePattern() (plan creation) and pattern() (CSR structure of Q, CSC structure of R generated on the device,
BlockDiagonalSparseQR.h:455-500,530-541) of the block-diagonal solver.  Usage (GPU box): python tools/pattern_probe.py"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import qrkit_amd as qa
ctx = qa.Context(0)
for (r, c, B) in ((7, 2, 256), (7, 2, 1000000), (8, 6, 20000), (32, 32, 10000), (64, 64, 20000)):
    rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
    tiles = torch.rand(B * r * c, device="cuda", dtype=torch.float64)
    mat = qa.SparseBlockDiagonal.fromTiles(rows, cols, tiles)
    qr = qa.BlockDiagonalSparseQR(context=ctx)
    t0 = time.perf_counter(); qr.analyzePattern(mat); torch.cuda.synchronize(); t1 = time.perf_counter()
    qr.factorize(mat); torch.cuda.synchronize()
    t2 = time.perf_counter(); p = qr.pattern(); torch.cuda.synchronize(); t3 = time.perf_counter()
    t4 = time.perf_counter(); p = qr.pattern(); torch.cuda.synchronize(); t5 = time.perf_counter()
    nnzq, nnzr = B * r * r, B * c * (c + 1) // 2
    byts = 4 * (nnzq + nnzr + B * r + B * c)
    print(f"{r:3d}x{c:<3d} B={B:8d}  analyzePattern {1e3*(t1-t0):8.2f} ms   pattern() first {1e3*(t3-t2):8.2f} ms, again {1e3*(t5-t4):8.2f} ms  ({byts/1e6:.0f} MB of indices)", flush=True)
