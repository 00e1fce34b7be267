"""Diagnostic: build the library with -DQRK_STAMP, run one 10000-tile factorisation and print where a
wave spends its cycles in every step (s_memtime deltas of lane 0 of one workgroup).  Never a timed build."""
import os, subprocess, sys, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "build", "libqrkit_amd_stamp.so")
srcs = [os.path.join(ROOT, "qrkit_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "qrkit_amd", "csrc")) if f.endswith(".hip")]
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-DQRK_STAMP"] + srcs + ["-o", out])
os.environ["QRKIT_AMD_LIB"] = out
import numpy as np, torch
import qrkit_amd
from qrkit_amd import _capi as capi
ctx = qrkit_amd.Context(0)
B = 10000
lay = capi.BDLayout(); lay.num_blocks, lay.block_rows, lay.block_cols = B, 32, 32; lay.rows = lay.cols = None; lay.mat_rows = lay.mat_cols = B * 32
plan = C.c_void_p(); capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), 0, 0, C.byref(plan)))
tiles = torch.rand(B * 1024, device="cuda", dtype=torch.float64) * 4.5 + 0.5
qv = torch.empty(B * 1024, device="cuda", dtype=torch.float64); rv = torch.empty(B * 528, device="cuda", dtype=torch.float64)
pm = torch.empty(B * 32, device="cuda", dtype=torch.int32)
stamps = torch.zeros(B * 32, device="cuda", dtype=torch.int64)
for _ in range(3):
    capi.check(capi.lib().qrk_bd_factorize(plan, tiles.data_ptr(), qv.data_ptr(), rv.data_ptr(), pm.data_ptr(), stamps.data_ptr(), 0))
torch.cuda.synchronize()
s = stamps[:32 * 8].cpu().numpy().reshape(32, 8)
names = ["dot", "scalars+park+downdate", "search+fetch(K+1)", "update", "refresh+rare"]
print("step   " + "  ".join(f"{n:>22s}" for n in names) + "   step_total   gap_to_next")
tot = np.zeros(5)
for k in range(32):
    d = [int(s[k, p + 1] - s[k, p]) for p in range(5)]
    gap = int(s[k + 1, 0] - s[k, 5]) if k < 31 else 0
    tot += d
    print(f"{k:4d}   " + "  ".join(f"{v:22d}" for v in d) + f"   {int(s[k,5]-s[k,0]):10d}   {gap:10d}")
print("sum    " + "  ".join(f"{int(v):22d}" for v in tot) + f"   {int(s[31,5]-s[0,0]):10d}")
