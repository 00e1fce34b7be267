"""Race detector for the multi-stream first stage of the two-stage dense QR: the arithmetic does not depend on the schedule, so the
packed result must be bitwise the same with the pipelined look-ahead (three streams), the plain look-ahead (two) and none (one),
run after run.  Usage (GPU box): python tools/caqr_race_check.py [repeats]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import qrkit_amd
from qrkit_amd.angular import DenseColPivQR

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = qrkit_amd.Context(0)
os.environ["QRK_DENSE_TWO_STAGE"] = "1"
os.environ["QRK_CAQR_NO_NARROW"] = "1"      # (the narrow kernel adds the chunks' partial sums in another order than the general one: with it the
                                            #  modes apply different kernels to the same columns and differ in the last bits, legitimately)
bad = 0
for rows, cols in ((40000, 2000), (20000, 1000), (33000, 700), (10000, 504), (8192, 256), (6000, 300), (4100, 200), (2048, 96)):
    g = torch.Generator(device="cuda"); g.manual_seed(rows + cols)
    A0 = torch.rand((cols, rows), device="cuda", dtype=torch.float64, generator=g) * 4.5 + 0.5
    ref = None
    for mode in ("pipe", "plain", "none"):
        os.environ["QRK_CAQR_PIPE"] = "1" if mode == "pipe" else "0"
        os.environ["QRK_CAQR_LOOKAHEAD"] = "0" if mode == "none" else "1"
        qr = DenseColPivQR(ctx, 0)                      # (the switches are read when the plan is created)
        for r in range(reps if mode == "pipe" else 1):
            At = A0.clone().t()
            qr.compute(At)
            torch.cuda.synchronize()
            res = (At.clone(), qr._hc.clone(), qr.colsPermutation().clone())
            if ref is None:
                ref = res
            else:
                same = all(torch.equal(a, b) for a, b in zip(res, ref))
                if not same:
                    bad += 1
                    print(f"{rows} x {cols}: {mode} run {r} differs from the first pipelined run", flush=True)
    print(f"{rows} x {cols}: {reps} pipelined runs, plain look-ahead and no look-ahead agree bitwise" if not bad else f"{rows} x {cols}: MISMATCH", flush=True)
sys.exit(1 if bad else 0)
