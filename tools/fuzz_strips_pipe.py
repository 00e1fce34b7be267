"""The pipelined strips chain (banded.hip, BBPipe: blocks of 16 columns, the carry handed over in two steps) run again and again on the same
input: a race between the workgroups would show as a factor that differs from run to run.  Every shape: R of 12 runs bitwise equal, equal to
rounding to the one-workgroup chain (QRK_BBS_PIPE=1: 32-column blocks, another order of operations), and the least-squares solution right.
Usage (GPU box): python tools/fuzz_strips_pipe.py [runs]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import qrkit_amd
from qrkit_amd.banded import BandedStripsQR

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ctx = qrkit_amd.Context(0)
shapes = [(300, 256, 192, 64), (257, 256, 192, 64), (400, 64, 48, 16), (333, 96, 64, 32), (150, 256, 256, 64), (200, 256, 240, 16),
          (120, 256, 192, 96), (500, 128, 128, 64), (64, 256, 192, 64), (3, 256, 192, 64)]
bad = 0
for (N, ms, n, s) in shapes:
    g = torch.Generator(device="cuda"); g.manual_seed(N * 7 + n)
    strips = torch.rand(N * ms * n, device="cuda", dtype=torch.float64, generator=g) * 2 - 1
    b = torch.rand(N * ms, device="cuda", dtype=torch.float64, generator=g)

    def rows_of(qr):
        return torch.cat([qr.rRows(i).reshape(-1) for i in range(0, N, max(1, N // 40))])

    os.environ["QRK_BBS_PIPE"] = "1"
    q1 = BandedStripsQR(N, ms, n, s, context=ctx); q1.factorize(strips)
    R1, x1 = rows_of(q1), q1.solve(b)
    os.environ.pop("QRK_BBS_PIPE")
    qp = BandedStripsQR(N, ms, n, s, context=ctx)
    first = None
    for r in range(runs):
        qp.factorize(strips)
        Rp = rows_of(qp)
        if first is None:
            first = Rp.clone()
        elif not torch.equal(first, Rp):
            bad += 1
            print(f"  {N} x ({ms} x {n}, step {s}): run {r} differs from run 0 in {int((first != Rp).sum())} entries")
    xp = qp.solve(b)
    # R up to the sign of each row: compare |R|
    dR = float((first.abs() - R1.abs()).norm() / R1.norm())
    dx = float((xp - x1).norm() / x1.norm())
    ok = dR <= 1e-11 and dx <= 1e-9
    bad += 0 if ok else 1
    print(f"{N:4d} strips of {ms} x {n}, step {s}: {runs} pipelined runs bitwise equal; against one workgroup |R| {dR:.1e}, x {dx:.1e}  {'ok' if ok else 'FAILED'}", flush=True)
print("fuzz_strips_pipe:", "all ok" if bad == 0 else f"{bad} FAILURES")
sys.exit(1 if bad else 0)
