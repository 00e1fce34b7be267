"""A stream of independent 10 000-tile matrices (BASELINE configs[1]) factorised through ONE handle / stream (every launch waits for the
previous one to drain: the 20-us chain of the last pairs is paid per launch) against TWO handles on two HIP streams, alternating (the
tail of one launch runs beside the head of the next).  Distinct input / output buffers per matrix in flight.  Wall clock around K steps.
Usage (GPU box): python tools/two_stream_probe.py [B] [K]"""
import ctypes as C, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import qrkit_amd
from qrkit_amd import _capi as capi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 400
S = 4                      # matrices rotated
dev = torch.device("cuda", 0)
lib = capi.lib()


def make(nstreams):
    out = []
    for i in range(nstreams):
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            ctx = qrkit_amd.Context(0)
        lay = capi.BDLayout()
        lay.num_blocks, lay.block_rows, lay.block_cols = B, 32, 32
        lay.rows = lay.cols = None
        lay.mat_rows = lay.mat_cols = B * 32
        plan = C.c_void_p()
        capi.check(lib.qrk_bd_plan_create(ctx.handle, C.byref(lay), capi.FULL_Q, capi.COLPIV_HOUSEHOLDER, C.byref(plan)), ctx.handle)
        out.append((st, ctx, plan))
    return out


g = torch.Generator(device=dev); g.manual_seed(1)
tiles = torch.rand(S * B * 1024, generator=g, device=dev, dtype=torch.float64) * 4.5 + 0.5
qv = torch.empty(S * B * 1024, device=dev, dtype=torch.float64)
rv = torch.empty(S * B * 528, device=dev, dtype=torch.float64)
pm = torch.empty(S * B * 32, device=dev, dtype=torch.int32)
torch.cuda.synchronize()


def step(lane, s):
    _, ctx, plan = lane
    capi.check(lib.qrk_bd_factorize(plan, tiles.data_ptr() + 8 * s * B * 1024, qv.data_ptr() + 8 * s * B * 1024, rv.data_ptr() + 8 * s * B * 528,
                                    pm.data_ptr() + 4 * s * B * 32, None, capi.MEM_DEVICE), ctx.handle)


for n in (1, 2, 1, 2, 3, 4):
    lanes = make(n)
    for it in range(20):
        step(lanes[it % n], it % S)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        for it in range(K):
            step(lanes[it % n], it % S)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K)
    ts.sort()
    us = ts[len(ts) // 2] * 1e6
    print(f"{n} stream(s): {us:7.2f} us per step (median of 5 x {K})   {1e6 / us:9.1f} factorizations/s   {B * 20736 / us / 1e3 / 8000:.3f} of 8 TB/s", flush=True)
    for _, ctx, plan in lanes:
        lib.qrk_bd_plan_destroy(plan)
