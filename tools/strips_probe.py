"""Banded strips form (qrk_bbs_*): ms per strip of factorize / solve at the BASELINE configs[2] strip shape.
Usage (GPU box): python tools/strips_probe.py [N ...]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import qrkit_amd
from qrkit_amd.banded import BandedStripsQR
ctx = qrkit_amd.Context(0)
ms, n, s = 256, 192, 64
for N in [int(a) for a in sys.argv[1:]] or [512, 2048]:
    strips = torch.rand(N * ms * n, device="cuda", dtype=torch.float64) * 2 - 1
    qr = BandedStripsQR(N, ms, n, s, context=ctx)
    qr.factorize(strips); torch.cuda.synchronize()
    t0 = time.perf_counter(); qr.factorize(strips); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    b = torch.rand(qr.rows(), device="cuda", dtype=torch.float64)
    qr.solve(b); torch.cuda.synchronize()
    t0 = time.perf_counter(); qr.solve(b); torch.cuda.synchronize(); ds = time.perf_counter() - t0
    print(f"strips form {N} x ({ms} x {n}, step {s}): factorize {dt * 1e3:9.2f} ms = {dt / N * 1e3:7.4f} ms per strip; solve (1 rhs) {ds * 1e3:8.2f} ms = {ds / N * 1e3:7.4f} ms per strip", flush=True)
