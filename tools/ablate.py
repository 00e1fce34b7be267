"""Diagnostic: what does each phase of the pair kernel's step cost?

  python tools/ablate.py build   (here)    compiles bdqr_pair.hip with -DQRK_ABL=<mask> for a list of masks
                                            into tools/abl/libqrk_abl_<mask>.so (links the other objects of build/obj)
  python tools/ablate.py run     (GPU box) times 10000 32x32 tiles with each library

Results with a mask != 0 are numerically wrong; only the time is of interest.
"""
import glob
import os
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MASKS = [int(x) for x in os.environ.get('QRK_MASKS', '0,1024,16,48,1+2+4+64+128+256'.replace('1+2+4+64+128+256', '455')).split(',')]
NAMES = {1: "no tie branch", 2: "no bpermute", 4: "no degenerate branch", 8: "no sqrt/recip", 16: "no update FMAs",
         32: "no dot FMAs", 2048: "no Q (A columns only)", 64: "no refresh/parking", 128: "no norm downdate", 256: "no x corrections", 1024: "no steps at all"}


def build():
    out = os.path.join(ROOT, "tools", "abl")
    os.makedirs(out, exist_ok=True)
    objs = [o for o in glob.glob(os.path.join(ROOT, "build", "obj", "*.o")) if not o.endswith("bdqr_pair.o")]
    procs = []
    for m in MASKS:
        o = os.path.join(out, f"pair_{m}.o")
        procs.append((m, o, subprocess.Popen(
            ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-DQRK_ABL={m}",
             "-I" + os.path.join(ROOT, "include"), "-c", os.path.join(ROOT, "qrkit_amd", "csrc", "bdqr_pair.hip"), "-o", o])))
        if len(procs) % 4 == 0:
            for _, _, p in procs[-4:]:
                p.wait()
    for m, o, p in procs:
        assert p.wait() == 0
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", o] + objs +
                              ["-o", os.path.join(out, f"libqrk_abl_{m}.so")])
        os.remove(o)


def run_one():
    import ctypes as C
    import torch
    sys.path.insert(0, ROOT)
    import qrkit_amd
    from qrkit_amd import _capi as capi
    ctx = qrkit_amd.Context(0)
    B = 10000
    lay = capi.BDLayout()
    lay.num_blocks, lay.block_rows, lay.block_cols = B, 32, 32
    lay.rows = lay.cols = None
    lay.mat_rows = lay.mat_cols = B * 32
    plan = C.c_void_p()
    capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), 0, 0, C.byref(plan)))
    S = 8
    g = torch.Generator(device="cuda").manual_seed(1)
    tiles = torch.rand(S * B * 1024, device="cuda", dtype=torch.float64, generator=g) * 4.5 + 0.5
    qv = torch.empty(S * B * 1024, device="cuda", dtype=torch.float64)
    rv = torch.empty(S * B * 528, device="cuda", dtype=torch.float64)
    pm = torch.empty(S * B * 32, device="cuda", dtype=torch.int32)

    def run(it):
        ms = C.c_float()
        capi.check(capi.lib().qrk_bd_time_factorize(plan, tiles.data_ptr(), qv.data_ptr(), rv.data_ptr(), pm.data_ptr(),
                                                    S, it, C.byref(ms)))
        return ms.value
    run(40)
    print(f"{run(400) * 1e3:.1f}")


def run():
    for m in MASKS:
        lib = os.path.join(ROOT, "tools", "abl", f"libqrk_abl_{m}.so")
        env = dict(os.environ, QRKIT_AMD_LIB=lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, capture_output=True, text=True)
        what = " + ".join(NAMES[b] for b in NAMES if m & b) or "full step"
        print(f"mask {m:4d}  {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]:>8s} us   {what}", flush=True)


if __name__ == "__main__":
    {"build": build, "run": run, "one": run_one}[sys.argv[1]]()
