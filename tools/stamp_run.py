"""Diagnostic: build the library with -DQRK_STAMP, run 10000-tile factorisations and print where the
persistent pair kernel's waves spend their time (s_memtime of lane 0 at phase boundaries of every pair).
Never a timed build."""
import os, subprocess, sys, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out = os.environ.get("QRK_STAMP_LIB") or os.path.join(ROOT, "build", "libqrkit_amd_stamp.so")
srcs = [os.path.join(ROOT, "qrkit_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "qrkit_amd", "csrc")) if f.endswith(".hip")]
os.makedirs(os.path.dirname(out), exist_ok=True)
if not os.path.exists(out) or os.environ.get("QRK_REBUILD"):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-DQRK_STAMP",
                           "-I" + os.path.join(ROOT, "include")] + srcs + ["-o", out])
if len(sys.argv) > 1 and sys.argv[1] == "build":
    sys.exit(0)
os.environ["QRKIT_AMD_LIB"] = out
import numpy as np, torch
import qrkit_amd
from qrkit_amd import _capi as capi
ctx = qrkit_amd.Context(0)
B = int(os.environ.get("QRK_B", "10000"))
lay = capi.BDLayout(); lay.num_blocks, lay.block_rows, lay.block_cols = B, 32, 32; lay.rows = lay.cols = None; lay.mat_rows = lay.mat_cols = B * 32
plan = C.c_void_p(); capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), 0, 0, C.byref(plan)))
tiles = torch.rand(B * 1024, device="cuda", dtype=torch.float64) * 4.5 + 0.5
qv = torch.empty(B * 1024, device="cuda", dtype=torch.float64); rv = torch.empty(B * 528, device="cuda", dtype=torch.float64)
pm = torch.empty(B * 32, device="cuda", dtype=torch.int32)
NP = (B + 1) // 2
stamps = torch.zeros(NP * 20, device="cuda", dtype=torch.int64)
for _ in range(3):
    capi.check(capi.lib().qrk_bd_factorize(plan, tiles.data_ptr(), qv.data_ptr(), rv.data_ptr(), pm.data_ptr(), stamps.data_ptr(), 0))
torch.cuda.synchronize()
# duration of one launch of this (stamped) build by events, to convert s_memtime ticks: ticks of a workgroup's whole life / this
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    capi.check(capi.lib().qrk_bd_factorize(plan, tiles.data_ptr(), qv.data_ptr(), rv.data_ptr(), pm.data_ptr(), stamps.data_ptr(), 0))
e1.record(); torch.cuda.synchronize()
launch_us = e0.elapsed_time(e1) * 1e3 / 20
raw = stamps.cpu().numpy()
s = raw[:NP * 12].reshape(NP, 12)
tk = raw[NP * 12:].reshape(NP, 8)
t0 = s[:, 0].min()
rel = (s[:, :7] - t0).astype(np.float64)
MHZ = 100.0   # s_memtime ticks at 100 MHz on gfx950
print("s_memtime ticks are converted with %.0f MHz; kernel span = %.1f us" % (MHZ, rel.max() / MHZ))
names = ["stage", "steps0-7", "steps8-15", "steps16-23", "steps24-31", "epilogue"]
d = np.diff(rel, axis=1) / MHZ
print("phase durations (us): mean / p10 / p90")
for i, n in enumerate(names):
    print(f"  {n:12s} {d[:, i].mean():8.2f} {np.percentile(d[:, i], 10):8.2f} {np.percentile(d[:, i], 90):8.2f}")
print(f"  {'pair total':12s} {(rel[:, 6] - rel[:, 0]).mean() / MHZ:8.2f}")
wg = s[:, 8]
# round 4: the same in raw s_memtime ticks (= shader cycles on gfx950), by round of the persistent workgroup -- rounds 0 and 1 run two
# waves per SIMD, the tail round mostly one: what a pair costs when it has the SIMD to itself is the dependent chain of its steps
nwg0 = int(wg.max()) + 1
dt = np.diff((s[:, :7] - s[:, :1]).astype(np.float64), axis=1)
life = np.array([s[[p for p in range(NP) if p % nwg0 == w], :7].max() - s[[p for p in range(NP) if p % nwg0 == w], :7].min() for w in range(0, nwg0, 64)], dtype=np.float64)
print(f"one launch of the stamped build by events: {launch_us:.1f} us; ticks from a workgroup's first stamp to its last: mean {life.mean():.0f} max {life.max():.0f}"
      f" -> {life.max() / launch_us / 1e3:.2f} GHz if the longest-lived workgroup spans the launch")
print("cycles per pair by round (stage | steps 0-7 | 8-15 | 16-23 | 24-31 | epilogue | total), mean over the pairs of the round:")
for r in range((NP + nwg0 - 1) // nwg0):
    sel = np.arange(r * nwg0, min((r + 1) * nwg0, NP))
    print(f"  round {r} ({len(sel):5d} pairs): " + " ".join(f"{x:8.0f}" for x in dt[sel].mean(axis=0)) + f"   total {dt[sel].sum(axis=1).mean():8.0f}"
          f"   p10 {np.percentile(dt[sel].sum(axis=1), 10):8.0f}  p90 {np.percentile(dt[sel].sum(axis=1), 90):8.0f}")
if os.environ.get("QRK_PAIR_PERSIST") == "0":
    d0 = (s[:, :7] - s[:, :1]).astype(np.float64)
    print("one pair per workgroup: mean stamp offsets within the workgroup (ticks):", " ".join(f"{x:9.0f}" for x in d0.mean(axis=0)))
    sys.exit(0)
order = np.argsort(rel[:, 0])
print("start times of pairs (us) by round of their workgroup:")
nwg = int(wg.max()) + 1
for r in range(3):
    sel = [p for p in range(NP) if p // nwg == r]
    if sel:
        print(f"  round {r}: n={len(sel)} start mean {rel[sel, 0].mean() / MHZ:8.2f} min {rel[sel, 0].min() / MHZ:8.2f} max {rel[sel, 0].max() / MHZ:8.2f}"
              f"   end mean {rel[sel, 6].mean() / MHZ:8.2f} max {rel[sel, 6].max() / MHZ:8.2f}")
# one workgroup's timeline
for w in (0, 1, nwg - 1):
    ps = [p for p in range(NP) if p % nwg == w]
    print(f"workgroup {w}: " + " | ".join(" ".join(f"{x / MHZ:6.1f}" for x in rel[p]) for p in ps))
dk = np.diff(tk.astype(np.float64), axis=1)
nwg_ = int(s[:, 8].max()) + 1
print("inside one step (ticks), phases: search | image read+corrections | publish+dot | bpermute+sqrt start | recip+gammas | update | park/refresh+downdate")
for label, sel in (("all pairs", np.arange(NP)), ("tail round (1 wave/SIMD mostly)", np.arange(2 * nwg_, NP))):
    if len(sel):
        print(f"  {label:34s}" + " ".join(f"{x:8.0f}" for x in dk[sel].mean(axis=0)) + f"   total {dk[sel].sum(axis=1).mean():8.0f}")
xcc = (s[:, 7] >> 32) & 0xf
print("pairs per XCC:", np.bincount(xcc.astype(np.int64)))
# where do the workgroups of the tail round sit?  HW_ID (gfx9): wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
hw = s[:, 7] & 0xffffffff
cu_key = (xcc.astype(np.int64) << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
for r in range(3):
    sel = np.arange(r * nwg_, min((r + 1) * nwg_, NP))
    if len(sel):
        u, c = np.unique(cu_key[sel], return_counts=True)
        print(f"round {r}: {len(sel)} pairs on {len(u)} distinct CUs, per-CU min/mean/max = {c.min()}/{c.mean():.2f}/{c.max()}")
