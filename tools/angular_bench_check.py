"""Why does bench.py's configs3 leg read 65 ms when tools/angular_probe.py reads 48?  Times the leg alone, then after the other legs."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import qrkit_amd
import bench
ctx = qrkit_amd.Context(0)
dev = torch.device("cuda", 0)
r = bench.angular_config3(ctx, dev, torch, np)
print("alone:", r["compute_ms"], r["solve_ms"], flush=True)
if len(sys.argv) > 1:
    bench.strips_config2(ctx, dev, torch, np, 512)
    r = bench.angular_config3(ctx, dev, torch, np)
    print("after strips:", r["compute_ms"], r["solve_ms"], flush=True)
# the probe's way: numpy data, one warm-up, one measurement
rng = np.random.default_rng(5)
B, rr, c, m2 = 20000, 8, 6, 2000
tiles = rng.uniform(0.5, 5.0, B * rr * c)
left = qrkit_amd.SparseBlockDiagonal.fromTiles(np.full(B, rr, np.int32), np.full(B, c, np.int32), tiles)
J2 = rng.uniform(0.5, 5.0, (B * rr, m2))
ba = qrkit_amd.BlockAngularSparseQR(context=ctx)
mat = qrkit_amd.BlockMatrix1x2(left, torch.from_numpy(np.ascontiguousarray(J2.T)).cuda().t())
for _ in range(3):
    ba.compute(mat); torch.cuda.synchronize()
    t0 = time.perf_counter(); ba.compute(mat); torch.cuda.synchronize(); print("probe-style:", (time.perf_counter() - t0) * 1e3, flush=True)
# the same data as tensors on the device (what bench does)
tl = torch.from_numpy(tiles).cuda()
left2 = qrkit_amd.SparseBlockDiagonal.fromTiles(np.full(B, rr, np.int32), np.full(B, c, np.int32), tl)
mat2 = qrkit_amd.BlockMatrix1x2(left2, torch.from_numpy(np.ascontiguousarray(J2.T)).cuda().t())
ba.compute(mat2); torch.cuda.synchronize()
t0 = time.perf_counter(); ba.compute(mat2); torch.cuda.synchronize(); print("device tiles:", (time.perf_counter() - t0) * 1e3, flush=True)
