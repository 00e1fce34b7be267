"""Diagnostic: build the library with -DQRK_Q32_PROF (s_memtime ticks per phase of bdqr_quad32.hip's step, printed by workgroup 0) and run a
few batches.  Never a timed build.  Usage: python tools/q32_prof.py build (here) / python tools/q32_prof.py (GPU box)"""
import os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "build", "libqrkit_amd_q32prof%s.so" % os.environ.get("QRK_Q32_TAG", ""))
if len(sys.argv) > 1 and sys.argv[1] == "build":
    objs = [os.path.join(ROOT, "build", "obj", f) for f in os.listdir(os.path.join(ROOT, "build", "obj")) if f.endswith(".o") and f != "bdqr_quad32.o"]
    o = out[:-3] + ".o"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DQRK_Q32_PROF"] + os.environ.get("QRK_Q32_FLAGS", "").split() +
                          ["-I" + os.path.join(ROOT, "include"), "-c", os.path.join(ROOT, "qrkit_amd", "csrc", "bdqr_quad32.hip"), "-o", o])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", o] + objs + ["-ldl", "-o", out])
    os.remove(o)
    sys.exit(0)
os.environ["QRKIT_AMD_LIB"] = out
os.environ["QRK_K1_FORM"] = "quad32"
import numpy as np, torch
import qrkit_amd as qa
ctx = qa.Context(0)
for b in (4, 1024, 8192):
    rows = np.full(b, 32, np.int32)
    tiles = torch.rand(b * 1024, device="cuda", dtype=torch.float64) * 4.5 + 0.5
    mat = qa.SparseBlockDiagonal.fromTiles(rows, rows, tiles)
    qr = qa.BlockDiagonalSparseQR(blockSolver=0, qFormat=0, context=ctx)
    print(f"--- {b} tiles ({'two waves per SIMD' if b > 4096 else 'one wave per SIMD at most'})", flush=True)
    for _ in range(2):
        qr.compute(mat); torch.cuda.synchronize()
