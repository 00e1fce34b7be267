"""Time per launch of bdqr_pair4 (QRK_PAIR_V2=1) against the number of resident waves (QRK_PAIR_WGS): latency- or throughput-bound?"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import p4_check as pc  # noqa: E402
import qrkit_amd       # noqa: E402

ctx = qrkit_amd.Context(0)
for B in (10000,):
    for wgs in (2500, 2560, 2816, 3328, 4096):
        os.environ["QRK_PAIR_WGS"] = str(wgs)
        t = pc.timeit(ctx, B, True, True)
        print(f"B={B} waves={wgs} ({wgs // 1024}/SIMD): {t:.2f} us", flush=True)
    os.environ.pop("QRK_PAIR_WGS", None)
    print(f"B={B} K1: {pc.timeit(ctx, B, False, True):.2f} us", flush=True)
