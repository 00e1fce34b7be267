"""Kernel time (qrk_bd_time_factorize: HIP events around back-to-back launches, no host work per launch) of uniform batches of n x n tiles for
small and large B: where does the one-wave-per-tile kernel's launch sit against its chain?  Usage (GPU box): python tools/w64_small_batches.py"""
import os, sys, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import qrkit_amd
from qrkit_amd import _capi as capi
ctx = qrkit_amd.Context(0)
for n in (32, 33, 40, 44, 48, 56, 64):
    for B in (2, 256, 1024, 2000, 4096, 20000):
        lay = capi.BDLayout(); lay.num_blocks, lay.block_rows, lay.block_cols = B, n, n; lay.rows = lay.cols = None; lay.mat_rows = lay.mat_cols = B * n
        plan = C.c_void_p(); capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), 0, 0, C.byref(plan)))
        S = max(1, min(8, 40000 // B))
        t = torch.rand(S * B * n * n, device="cuda", dtype=torch.float64) * 2 - 1
        q = torch.empty(S * B * n * n, device="cuda", dtype=torch.float64); r = torch.empty(S * B * (n * (n + 1) // 2), device="cuda", dtype=torch.float64)
        p = torch.empty(S * B * n, device="cuda", dtype=torch.int32)
        ms = C.c_float()
        def run(it):
            capi.check(capi.lib().qrk_bd_time_factorize(plan, t.data_ptr(), q.data_ptr(), r.data_ptr(), p.data_ptr(), S, it, C.byref(ms)))
            return ms.value
        run(20)
        us = min(run(100), run(100)) * 1e3
        print(f"{n:3d} x {n:<3d} B={B:6d}  {us:9.2f} us per launch  {B / us:9.2f} M tiles/s", flush=True)
        capi.lib().qrk_bd_plan_destroy(plan)
