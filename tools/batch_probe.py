"""How much of a 10000-tile launch is tail effect?  Time launches of B tiles for several B (one stream)."""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qrkit_amd
from qrkit_amd import _capi as capi
ctx = qrkit_amd.Context(0)
for B in (2048, 4096, 10000, 20000, 40000, 160000):
    lay = capi.BDLayout(); lay.num_blocks, lay.block_rows, lay.block_cols = B, 32, 32; lay.rows = lay.cols = None; lay.mat_rows = lay.mat_cols = B * 32
    plan = C.c_void_p(); capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), 0, 0, C.byref(plan)))
    S = max(1, 80000 // B)
    tiles = torch.rand(S * B * 1024, device="cuda", dtype=torch.float64) * 4.5 + 0.5
    qv = torch.empty(S * B * 1024, device="cuda", dtype=torch.float64); rv = torch.empty(S * B * 528, device="cuda", dtype=torch.float64)
    pm = torch.empty(S * B * 32, device="cuda", dtype=torch.int32)
    ms = C.c_float()
    for it in (5, 40):
        capi.check(capi.lib().qrk_bd_time_factorize(plan, tiles.data_ptr(), qv.data_ptr(), rv.data_ptr(), pm.data_ptr(), S, it, C.byref(ms)))
    print(f"B={B:7d}  {ms.value*1e3:9.1f} us/launch  {ms.value*1e3/B*1e4:8.1f} us per 10k tiles  {20736*B/ms.value/1e6:8.1f} GB/s")
    capi.lib().qrk_bd_plan_destroy(plan)
    del tiles, qv, rv, pm
