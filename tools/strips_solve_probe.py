"""Banded strips form (qrk_bbs_*): solve() through the carry maps (banded_maps.hip) against the one-workgroup chains (QRK_BBS_MAPS=0),
first call (maps built) and steady state, plus Q^T b and Q x on their own.
Usage (GPU box): python tools/strips_solve_probe.py [N ...]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import qrkit_amd
from qrkit_amd.banded import BandedStripsQR
ctx = qrkit_amd.Context(0)
ms, n, s = 256, 192, 64


def timed(fn, reps=3):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts)


for N in [int(a) for a in sys.argv[1:]] or [2048]:
    strips = torch.rand(N * ms * n, device="cuda", dtype=torch.float64) * 2 - 1
    b = torch.rand(N * ms, device="cuda", dtype=torch.float64)
    res = {}
    for sw, kk in (("0", None), ("1", "0"), ("1", None)):
        os.environ["QRK_BBS_MAPS"] = sw
        os.environ.pop("QRK_BBS_MAPS_K", None)
        if kk is not None:
            os.environ["QRK_BBS_MAPS_K"] = kk
        qr = BandedStripsQR(N, ms, n, s, context=ctx)
        qr.factorize(strips); torch.cuda.synchronize()
        t0 = time.perf_counter(); x = qr.solve(b); torch.cuda.synchronize(); first = time.perf_counter() - t0
        t_solve = timed(lambda: qr.solve(b))
        t_qt = timed(lambda: qr.applyQ(b, transpose=True))
        y = qr.applyQ(b, transpose=True)
        t_q = timed(lambda: qr.applyQ(y, transpose=False))
        back = qr.applyQ(y, transpose=False)
        res[sw] = x
        print(f"N={N} QRK_BBS_MAPS={sw} K={kk if kk is not None else 'auto'}: solve first call {first * 1e3:8.2f} ms, steady {t_solve * 1e3:8.2f} ms = {t_solve / N * 1e6:7.2f} us per strip; "
              f"Q^T b {t_qt * 1e3:8.2f} ms, Q x {t_q * 1e3:8.2f} ms; |QQ^T b - b|/|b| = {float((back - b).norm() / b.norm()):.2e}", flush=True)
        del qr
    print(f"   solutions of the two forms differ by {float((res['1'] - res['0']).norm() / res['0'].norm()):.2e} (relative)", flush=True)
