"""Diagnostic: timeline of one launch of bdqr_quad32 (library built with -DQRK_Q32_STAMP: s_memrealtime, 100 MHz, of every quad at the
start of its round, when its tiles are in registers, at the end of phase 1 and at its end).  Never a timed build.
  python tools/p4_stamps.py build      (here)
  python tools/p4_stamps.py [B]        (GPU box)"""
import os, subprocess, sys, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "tools", "abl_stamp", "libqrk_q32stamp%s.so" % os.environ.get("QRK_Q32_TAG", ""))
if len(sys.argv) > 1 and sys.argv[1] == "build":
    import glob
    os.makedirs(os.path.dirname(out), exist_ok=True)
    objs = [o for o in glob.glob(os.path.join(ROOT, "build", "obj", "*.o")) if not o.endswith("bdqr_quad32.o")]
    o = out[:-3] + ".o"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DQRK_Q32_STAMP"] + os.environ.get("QRK_Q32_FLAGS", "").split() + ["-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "qrkit_amd", "csrc"), "-c", os.path.join(ROOT, "qrkit_amd", "csrc", "bdqr_quad32.hip"), "-o", o])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", o] + objs + ["-o", out])
    os.remove(o)
    sys.exit(0)
os.environ["QRKIT_AMD_LIB"] = out
os.environ["QRK_K1_FORM"] = "quad32"
import numpy as np, torch
import qrkit_amd
from qrkit_amd import _capi as capi
ctx = qrkit_amd.Context(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
lay = capi.BDLayout(); lay.num_blocks, lay.block_rows, lay.block_cols = B, 32, 32; lay.rows = lay.cols = None; lay.mat_rows = lay.mat_cols = B * 32
plan = C.c_void_p(); capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), 0, 0, C.byref(plan)))
S = 8
tiles = torch.rand(S * B * 1024, device="cuda", dtype=torch.float64) * 4.5 + 0.5
qv = torch.empty(S * B * 1024, device="cuda", dtype=torch.float64); rv = torch.empty(S * B * 528, device="cuda", dtype=torch.float64)
pm = torch.empty(B * 32, device="cuda", dtype=torch.int32)
NP = (B + 3) // 4
stamps = torch.zeros(NP * 4 + 64, device="cuda", dtype=torch.int64)
def launch(i):
    capi.check(capi.lib().qrk_bd_factorize(plan, tiles.data_ptr() + (i % S) * B * 8192, qv.data_ptr() + (i % S) * B * 8192, rv.data_ptr() + (i % S) * B * 4224,
                                           pm.data_ptr(), stamps.data_ptr(), capi.MEM_DEVICE))
for i in range(10):
    launch(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(40):
    launch(i)
e1.record(); torch.cuda.synchronize()
print(f"B={B}: {e0.elapsed_time(e1) * 1e3 / 40:.2f} us per launch (stamped build, 40 launches back to back)")
s = stamps.cpu().numpy()[:NP * 4].reshape(NP, 4).astype(np.float64) / 100.0       # us
t0 = s[:, 0].min()
s -= t0
G = min(NP, 2048)
print(f"span first stamp -> last stamp: {s.max():.2f} us")
def q(x): return f"min {x.min():6.2f}  p10 {np.percentile(x, 10):6.2f}  median {np.median(x):6.2f}  p90 {np.percentile(x, 90):6.2f}  max {x.max():6.2f}"
for name, idx in (("first quad of a wave", np.arange(0, min(NP, G))), ("second quad of a wave", np.arange(G, NP))):
    if len(idx) == 0: continue
    x = s[idx]
    print(f"--- {name}: {len(idx)} pairs")
    print("  round start      ", q(x[:, 0]))
    print("  tiles in regs    ", q(x[:, 1]))
    print("  load wait        ", q(x[:, 1] - x[:, 0]))
    print("  phase 1 (A -> R) ", q(x[:, 2] - x[:, 1]))
    print("  phase 2 (Q)      ", q(x[:, 3] - x[:, 2]))
    print("  quad end           ", q(x[:, 3]))
# by launch generation (blockIdx / 1024): do the first waves of a SIMD get their data first?
for g in range((G + 1023) // 1024):
    idx = np.arange(g * 1024, min((g + 1) * 1024, G))
    print(f"  waves {g * 1024}..: start median {np.median(s[idx, 0]):6.2f}  tiles-in-regs median {np.median(s[idx, 1]):6.2f}  end median {np.median(s[idx, 3]):6.2f} max {s[idx, 3].max():6.2f}")
# how many waves are computing at time t
ts = np.arange(0, s.max() + 2, 2.0)
busy = [(int(((s[:, 1] <= t) & (s[:, 3] > t)).sum()), int(((s[:, 0] <= t) & (s[:, 1] > t)).sum())) for t in ts]
print("t (us): waves computing / waves waiting for their tiles")
print("  " + "  ".join(f"{t:.0f}:{b[0]}/{b[1]}" for t, b in zip(ts, busy)))
