"""Kernel time (qrk_bd_time_factorize) of uniform batches of small tiles: the several-tiles-per-wave kernel (bdqr_quad.hip: 9..16 rows four per wave; with the argument `small` 5..8 rows, eight per wave) against
bdqr_small.hip's 16-lane groups (QRK_QUAD=0).  Usage (GPU box): python tools/quad_probe.py"""
import os, sys, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import qrkit_amd
from qrkit_amd import _capi as capi
ctx = qrkit_amd.Context(0)
SHAPES = ((16, 16), (12, 12), (9, 9), (16, 8), (12, 6), (10, 4)) if len(sys.argv) < 2 else ((8, 8), (8, 6), (6, 6), (7, 4), (5, 5), (8, 3))
for (r, c) in SHAPES:
    for B in ((20000, 400000) if r > 8 else (20000, 1000000)):
        lay = capi.BDLayout(); lay.num_blocks, lay.block_rows, lay.block_cols = B, r, c; lay.rows = lay.cols = None; lay.mat_rows, lay.mat_cols = B * r, B * c
        plan = C.c_void_p(); capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), 0, 0, C.byref(plan)))
        S = max(1, min(8, 800000 // B))
        t = torch.rand(S * B * r * c, device="cuda", dtype=torch.float64) * 2 - 1
        q = torch.empty(S * B * r * r, device="cuda", dtype=torch.float64); rv = torch.empty(S * B * (c * (c + 1) // 2), device="cuda", dtype=torch.float64)
        p = torch.empty(S * B * c, device="cuda", dtype=torch.int32)
        ms = C.c_float()
        def run(it):
            capi.check(capi.lib().qrk_bd_time_factorize(plan, t.data_ptr(), q.data_ptr(), rv.data_ptr(), p.data_ptr(), S, it, C.byref(ms)))
            return ms.value
        run(10)
        us = min(run(50), run(50)) * 1e3
        by = 8 * r * c + 8 * r * r + 4 * c * (c + 1) + 4 * c
        print(f"{r:3d} x {c:<3d} B={B:7d}  {us:9.2f} us per launch  {B / us:9.2f} M tiles/s  {B * by / us / 1e3:8.1f} GB/s = {B * by / us / 8e6:5.3f} of HBM", flush=True)
        capi.lib().qrk_bd_plan_destroy(plan)
