"""The mixed 8...256 batch of tools/mixed_probe.py alone (for rocprofv3 --kernel-trace).  Usage: python tools/mixed_only.py [B] [lo] [hi]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import qrkit_amd as qa

ctx = qa.Context(0)
rng = np.random.default_rng(1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 8
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 256
n = rng.integers(lo, hi + 1, B).astype(np.int32)
tiles = torch.rand(int((n.astype(np.int64) ** 2).sum()), device="cuda", dtype=torch.float64) * 2 - 1
mat = qa.SparseBlockDiagonal.fromTiles(n, n, tiles)
qr = qa.BlockDiagonalSparseQR(blockSolver=0, qFormat=0, context=ctx)
qr.analyzePattern(mat)
qr.factorize(mat); torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    qr.factorize(mat); torch.cuda.synchronize()
    print(f"mixed {lo}..{hi} B={B}: {(time.perf_counter() - t0) * 1e3:.3f} ms", flush=True)
