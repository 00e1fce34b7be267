"""Time of the dense exact path over the whole chip on a block whose pivot decisions are ties (+-1 entries): tall right block of
BASELINE configs[3] and smaller ones.  Usage (GPU box): python tools/exact_wide_probe.py [rows cols]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("QRK_DEBUG_UNCLEAR", "1")
import numpy as np, torch
import qrkit_amd
from qrkit_amd.angular import DenseColPivQR

shapes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(4096, 256), (10000, 504), (40000, 2000)]
ctx = qrkit_amd.Context(0)
for rows, cols in shapes:
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    A0 = (torch.randint(0, 2, (cols, rows), device="cuda", generator=g, dtype=torch.int32) * 2 - 1).to(torch.float64)
    qr = DenseColPivQR(ctx, 0)
    A = A0.clone().t()
    qr.compute(A); torch.cuda.synchronize()
    A = A0.clone().t()
    t0 = time.perf_counter(); qr.compute(A); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    R = torch.triu(A[:cols, :])
    P = qr.colsPermutation().long()
    # A P = Q R: |R^T R - (A P)^T (A P)| small
    AP = A0.t()[:, P]
    err = (R.t() @ R - AP.t() @ AP).abs().max().item() / rows
    print(f"{rows} x {cols} (+-1 entries): factorize incl. the exact path {dt * 1e3:.1f} ms, |R^T R - (AP)^T AP| / rows = {err:.2e}", flush=True)
