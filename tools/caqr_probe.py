"""Timing probe for the dense right-block solver on tall matrices: direct level-2 path vs the two-stage form (caqr.hip).
Usage (GPU box): python tools/caqr_probe.py [rows cols]      QRK_DENSE_TWO_STAGE=0/1 selects the form"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import qrkit_amd as qa
from qrkit_amd.angular import DenseColPivQR

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
ctx = qa.Context(0)
g = torch.Generator(device="cuda").manual_seed(3)
A0 = (torch.rand((cols, rows), device="cuda", dtype=torch.float64, generator=g) * 4.5 + 0.5)
qr = DenseColPivQR(ctx, 0)
for it in range(3):
    At = A0.clone().t()          # column-major rows x cols
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    qr.compute(At)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{rows} x {cols}  two_stage={os.environ.get('QRK_DENSE_TWO_STAGE', 'auto')}  factorize {dt * 1e3:8.2f} ms", flush=True)
