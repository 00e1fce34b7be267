"""Diagnostic A/B harness for the pair kernel (K1).

  python tools/ab.py build name1:-DFOO=1 name2:-DBAR=2 ...   (here)  bdqr_pair.hip compiled with the given flags into
                                                                      tools/abl/libqrk_<name>.so (other objects from build/obj)
  python tools/ab.py run [B]                                  (GPU)  times B (default 10000) 32x32 tiles with every library
                                                                      in tools/abl, three interleaved passes; us per launch
Only the time is of interest; variants may be numerically wrong.
"""
import glob
import os
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tools", "abl")


def build(specs):
    os.makedirs(OUT, exist_ok=True)
    for f in glob.glob(os.path.join(OUT, "*.so")):
        os.remove(f)
    base = os.environ.get("QRK_AB_SRC", "bdqr_pair.hip")      # which kernel source the variants are builds of
    objs = [o for o in glob.glob(os.path.join(ROOT, "build", "obj", "*.o")) if not o.endswith(base.replace(".hip", ".o"))]
    procs = []
    for spec in specs:
        name, _, flags = spec.partition(":")
        src = os.path.join(ROOT, "qrkit_amd", "csrc", base)
        if flags.startswith("@"):            # name:@path -> another source file
            src, flags = flags[1:], ""
        o = os.path.join(OUT, f"pair_{name}.o")
        procs.append((name, o, subprocess.Popen(
            ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
             "-I" + os.path.join(ROOT, "qrkit_amd", "csrc")] + flags.split() + ["-c", "-x", "hip", src, "-o", o])))
        if len(procs) % 6 == 0:
            for _, _, p in procs[-6:]:
                p.wait()
    for name, o, p in procs:
        assert p.wait() == 0, name
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", o] + objs +
                              ["-o", os.path.join(OUT, f"libqrk_{name}.so")])
        os.remove(o)


def run_one(B):
    import ctypes as C
    import torch
    sys.path.insert(0, ROOT)
    import qrkit_amd
    from qrkit_amd import _capi as capi
    ctx = qrkit_amd.Context(0)
    lay = capi.BDLayout()
    lay.num_blocks, lay.block_rows, lay.block_cols = B, 32, 32
    lay.rows = lay.cols = None
    lay.mat_rows = lay.mat_cols = B * 32
    plan = C.c_void_p()
    capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), 0, 0, C.byref(plan)))
    S = max(1, min(8, (80000 + B - 1) // B))
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
    run(30)
    if os.environ.get("QRK_AB_HASH"):
        # one factorisation of the first matrix through the product entry point; digest of perm / R / Q (bitwise comparison of builds)
        import hashlib
        capi.check(capi.lib().qrk_bd_factorize(plan, tiles.data_ptr(), qv.data_ptr(), rv.data_ptr(), pm.data_ptr(), None, capi.MEM_DEVICE))
        torch.cuda.synchronize()
        dig = [hashlib.sha1(x.cpu().numpy().tobytes()).hexdigest()[:10] for x in (pm[:B * 32], rv[:B * 528], qv[:B * 1024])]
        print("HASH perm=%s R=%s Q=%s" % tuple(dig))
    print(f"{run(300) * 1e3:.2f}")


def run(B):
    libs = sorted(glob.glob(os.path.join(OUT, "libqrk_*.so")))
    res = {l: [] for l in libs}
    for _ in range(3):
        for lib in libs:
            env = dict(os.environ, QRKIT_AMD_LIB=lib)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "one", str(B)], env=env, capture_output=True, text=True)
            out = r.stdout.strip().splitlines()
            for ln in out:
                if ln.startswith("HASH"):
                    print(os.path.basename(lib)[7:-3], ln, flush=True)
            res[lib].append(float(out[-1]) if out and out[-1].replace(".", "").isdigit() else float("nan"))
            if r.returncode != 0:
                print(os.path.basename(lib), "FAILED", r.stderr[-400:], flush=True)
    for lib in libs:
        v = res[lib]
        print(f"{os.path.basename(lib)[7:-3]:24s} " + " ".join(f"{x:8.2f}" for x in v) + f"   min {min(v):8.2f} us  (B={B})", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    elif sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 10000)
    else:
        run_one(int(sys.argv[2]))
