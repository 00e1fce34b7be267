"""BASELINE configs[3] on one GPU: compute() once, then solve() five times -- to be run under rocprofv3 --kernel-trace --stats so that the
kernels of the SOLVE path show up with their calls and durations (tools/run_r5_angsolve.sh)."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import qrkit_amd
B, r, c, m2 = 20000, 8, 6, 2000
dev = torch.device("cuda", 0)
ctx = qrkit_amd.Context(0)
g = torch.Generator(device=dev); g.manual_seed(778)
tl = torch.rand(B * r * c, generator=g, device=dev, dtype=torch.float64) * 4.5 + 0.5
left = qrkit_amd.SparseBlockDiagonal.fromTiles(np.full(B, r, np.int32), np.full(B, c, np.int32), tl)
J2 = (torch.rand(m2, B * r, generator=g, device=dev, dtype=torch.float64) * 4.5 + 0.5).t()
ba = qrkit_amd.BlockAngularSparseQR(context=ctx)
ba.compute(qrkit_amd.BlockMatrix1x2(left, J2)); torch.cuda.synchronize()
b = torch.rand(B * r, generator=g, device=dev, dtype=torch.float64)
ba.solve(b); torch.cuda.synchronize()
print("MARK solve start", flush=True)
t0 = time.perf_counter()
for _ in range(5):
    ba.solve(b)
torch.cuda.synchronize()
print(f"solve {1e3 * (time.perf_counter() - t0) / 5:.2f} ms", flush=True)
