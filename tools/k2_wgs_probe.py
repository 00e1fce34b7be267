"""K2 probe: tiles/s of uniform large tiles against the number of workgroups in flight (QRK_COL_WGS): is the level-2 pass served
faster when the working set of the resident workgroups fits L2 (4 MB per XCD)?  Usage (GPU box): python tools/k2_wgs_probe.py"""
import os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    sys.path.insert(0, ROOT)
    import time
    import numpy as np, torch
    import qrkit_amd as qa
    ctx = qa.Context(0)
    for s, b in ((256, 1024), (224, 1024), (192, 1024), (160, 1024), (128, 2048), (96, 2048)):
        rows = np.full(b, s, np.int32)
        tiles = torch.rand(b * s * s, device="cuda", dtype=torch.float64) * 2 - 1
        mat = qa.SparseBlockDiagonal.fromTiles(rows, rows, tiles)
        qr = qa.BlockDiagonalSparseQR(blockSolver=0, qFormat=0, context=ctx)
        qr.analyzePattern(mat)
        qr.factorize(mat); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            qr.factorize(mat)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        print(f"  {s:3d}x{s:<3d} B={b:5d} {dt*1e3:9.3f} ms {b/dt:10.0f} tiles/s {8*s**3/3*b/dt/1e12:6.2f} TFLOP/s (8 n^3/3 per tile, SURVEY.md 8(d))", flush=True)
    sys.exit(0)
for w in (sys.argv[1:] or ["0", "48", "64", "96", "128", "192", "256", "384"]):
    env = dict(os.environ)
    if w != "0":
        env["QRK_COL_WGS"] = w
    print(f"QRK_COL_WGS={w} (0 = the plan's own choice)", flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env)
