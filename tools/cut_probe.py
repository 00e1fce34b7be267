"""Time of qrk_bd_tiles_from_sparse (SparseBlockDiagonal::fromBlockDiagonalPattern, SparseBlockDiagonal.h:71-89) on the device: a CSC / CSR
matrix with a block-diagonal pattern of r x c blocks cut into dense tiles.  Kernel time by HIP events around repeated calls (index arrays
resident).  Usage (GPU box): python tools/cut_probe.py"""
import os, sys, time, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import qrkit_amd as qa
from qrkit_amd import _capi as capi
ctx = qa.Context(0)
dev = ctx.device
for (r, c, B) in ((7, 2, 256), (7, 2, 200000), (8, 6, 20000), (32, 32, 10000), (64, 64, 2000)):
    for fmt in ("csc", "csr"):
        # block-diagonal pattern, every entry of a block present
        nnz = B * r * c
        vals = torch.rand(nnz, device=dev, dtype=torch.float64)
        if fmt == "csc":
            ptr = torch.arange(0, B * c + 1, device=dev, dtype=torch.int32) * r
            idx = (torch.arange(nnz, device=dev, dtype=torch.int64) % r + (torch.arange(nnz, device=dev, dtype=torch.int64) // (r * c)) * r).to(torch.int32)
        else:
            ptr = torch.arange(0, B * r + 1, device=dev, dtype=torch.int32) * c
            idx = (torch.arange(nnz, device=dev, dtype=torch.int64) % c + (torch.arange(nnz, device=dev, dtype=torch.int64) // (r * c)) * c).to(torch.int32)
        lay = capi.BDLayout(); lay.num_blocks, lay.block_rows, lay.block_cols = B, r, c; lay.rows = lay.cols = None; lay.mat_rows, lay.mat_cols = B * r, B * c
        plan = C.c_void_p(); capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), 0, 0, C.byref(plan)), ctx.handle)
        tiles = torch.empty(nnz, device=dev, dtype=torch.float64)
        def once():
            capi.check(capi.lib().qrk_bd_tiles_from_sparse(plan, 1 if fmt == "csr" else 0, ptr.data_ptr(), idx.data_ptr(), vals.data_ptr(), nnz, tiles.data_ptr(), capi.MEM_DEVICE), ctx.handle)
        once(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): once()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        byts = nnz * (8 + 4 + 8)
        print(f"{r:3d}x{c:<3d} B={B:7d} {fmt}: {dt*1e6:9.1f} us  {byts/dt/1e9:8.1f} GB/s ({byts/dt/8e12*100:4.1f} % of 8 TB/s: values + inner indices read, tiles written)", flush=True)
        capi.lib().qrk_bd_plan_destroy(plan)
