"""Diagnostic: build the library with -DQRK_W64_PROF (s_memtime ticks per phase of bdqr_w64.hip's step, printed by workgroup 0) and run a
few batches.  Never a timed build.  Usage: python tools/w64_prof.py build (here) / python tools/w64_prof.py (GPU box)"""
import os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "build", "libqrkit_amd_w64prof.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    objs = [os.path.join(ROOT, "build", "obj", f) for f in os.listdir(os.path.join(ROOT, "build", "obj")) if f.endswith(".o") and f != "bdqr_w64.o"]
    o = os.path.join(ROOT, "build", "w64prof.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DQRK_W64_PROF", "-I" + os.path.join(ROOT, "include"),
                           "-c", os.path.join(ROOT, "qrkit_amd", "csrc", "bdqr_w64.hip"), "-o", o])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", o] + objs + ["-ldl", "-o", out])
    sys.exit(0)
os.environ["QRKIT_AMD_LIB"] = out
import numpy as np, torch
import qrkit_amd as qa
ctx = qa.Context(0)
for s, b in ((33, 20000), (64, 20000), (64, 256)):
    rows = np.full(b, s, np.int32)
    tiles = torch.rand(b * s * s, device="cuda", dtype=torch.float64) * 2 - 1
    mat = qa.SparseBlockDiagonal.fromTiles(rows, rows, tiles)
    qr = qa.BlockDiagonalSparseQR(blockSolver=0, qFormat=0, context=ctx)
    print(f"--- {s} x {s}, {b} tiles ({'two waves per SIMD' if b > 1024 else 'one wave per SIMD at most'})", flush=True)
    qr.compute(mat); torch.cuda.synchronize()
