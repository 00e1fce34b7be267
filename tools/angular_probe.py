"""Timing probe for the block-angular composition at the reference's test size (test/test-qrkit.cpp:386-395:
1024 variables -> 1024 tiles of 7x2, 384 dense columns) and at a BASELINE configs[3]-like left part.
Usage (GPU box): python tools/angular_probe.py"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import qrkit_amd as qa

ctx = qa.Context(0)
rng = np.random.default_rng(5)


def run(B, r, c, m2, label):
    tiles = rng.uniform(0.5, 5.0, B * r * c)
    left = qa.SparseBlockDiagonal.fromTiles(np.full(B, r, np.int32), np.full(B, c, np.int32), tiles)
    J2 = rng.uniform(0.5, 5.0, (B * r, m2))
    ba = qa.BlockAngularSparseQR(context=ctx)
    mat = qa.BlockMatrix1x2(left, torch.from_numpy(np.ascontiguousarray(J2.T)).cuda().t())   # column-major on the device, as Eigen's MatrixXd
    ba.compute(mat); torch.cuda.synchronize()
    t0 = time.perf_counter(); ba.compute(mat); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    x = rng.uniform(-1, 1, B * c + m2)
    # b = [J1 | J2] x
    b = J2 @ x[B * c:]
    T = tiles.reshape(B, c, r)
    b += np.einsum("bcr,bc->br", T, x[:B * c].reshape(B, c)).reshape(-1)
    xs = ba.solve(b)
    print(f"{label:46s} compute {dt*1e3:9.1f} ms   LS recovery {np.linalg.norm(xs - x) / np.linalg.norm(x):.2e}", flush=True)


run(1024, 7, 2, 384, "reference test size: 1024 x (7x2) + 384 dense")
run(2000, 8, 6, 200, "2000 x (8x6) + 200 dense")
run(20000, 8, 6, 200, "20000 x (8x6) + 200 dense")
if os.environ.get("QRK_BIG"):
    run(20000, 8, 6, 2000, "BASELINE configs[3] shape: 20000 x (8x6) + 2000 dense")
