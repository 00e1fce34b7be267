import os, sys, subprocess
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    sys.path.insert(0, ROOT)
    import numpy as np, torch, ctypes as C
    import qrkit_amd
    from qrkit_amd import _capi as capi
    B = int(sys.argv[2])
    ctx = qrkit_amd.Context(0)
    lay = capi.BDLayout(); lay.num_blocks, lay.block_rows, lay.block_cols = B, 32, 32; lay.rows = lay.cols = None; lay.mat_rows = lay.mat_cols = B * 32
    plan = C.c_void_p(); capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), 0, 0, C.byref(plan)))
    g = torch.Generator(device="cuda").manual_seed(1)
    tiles = torch.rand(B * 1024, device="cuda", dtype=torch.float64, generator=g) * 4.5 + 0.5
    qv = torch.zeros(B * 1024, device="cuda", dtype=torch.float64); rv = torch.empty(B * 528, device="cuda", dtype=torch.float64); pm = torch.empty(B * 32, device="cuda", dtype=torch.int32)
    capi.check(capi.lib().qrk_bd_factorize(plan, tiles.data_ptr(), qv.data_ptr(), rv.data_ptr(), pm.data_ptr(), None, 0)); torch.cuda.synchronize()
    np.save(sys.argv[3], qv.cpu().numpy())
    sys.exit(0)
import numpy as np
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
for name, env in (("/tmp/q_pc.npy", {}), ("/tmp/q_ref.npy", {"QRK_K1_PC": "0"})):
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "one", str(B), name], env=dict(os.environ, **env))
a = np.load("/tmp/q_pc.npy").reshape(B, 32, 32); b = np.load("/tmp/q_ref.npy").reshape(B, 32, 32)
bad = np.nonzero(np.any(a != b, axis=(1, 2)))[0]
print("tiles", B, "differing", len(bad), "first", bad[:20])
if len(bad):
    t = bad[0]
    d = a[t] != b[t]
    print("tile", t, "pair", t // 2, "wg", (t // 2) % 2048, "round", (t // 2) // 2048, "rows with diffs", np.nonzero(d.any(axis=1))[0][:32], "cols", np.nonzero(d.any(axis=0))[0][:32])
    print("max abs diff", np.abs(a[t] - b[t]).max(), "nan?", np.isnan(a[t]).any())
    rounds = (bad // 2) // 2048
    print("bad tiles per round:", np.bincount(rounds))
    print("bad halves:", np.bincount(bad % 2))
