"""Diagnostic: the two two-phase 32 x 32 kernels side by side in one process (QRK_K1_FORM is read when a plan is created).
  python tools/q32_ab.py [B ...]      (GPU box)   us per launch of bdqr_pair4 / bdqr_quad32 (staged and direct loads), interleaved passes;
                                                    perm / R / Q of one factorisation compared bitwise."""
import ctypes as C
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import qrkit_amd
from qrkit_amd import _capi as capi

ctx = qrkit_amd.Context(0)


def make_plan(B, form, direct=None):
    os.environ["QRK_K1_FORM"] = form
    if direct is None:
        os.environ.pop("QRK_Q32_DIRECT", None)
    lay = capi.BDLayout()
    lay.num_blocks, lay.block_rows, lay.block_cols = B, 32, 32
    lay.rows = lay.cols = None
    lay.mat_rows = lay.mat_cols = B * 32
    plan = C.c_void_p()
    capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), 0, 0, C.byref(plan)))
    return plan


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [1250, 2500, 5000, 8192, 10000, 20000, 100000]
    for B in sizes:
        S = max(1, min(8, (80000 + B - 1) // B))
        g = torch.Generator(device="cuda").manual_seed(1)
        tiles = torch.rand(S * B * 1024, device="cuda", dtype=torch.float64, generator=g) * 4.5 + 0.5
        qv = torch.empty(S * B * 1024, device="cuda", dtype=torch.float64)
        rv = torch.empty(S * B * 528, device="cuda", dtype=torch.float64)
        pm = torch.empty(S * B * 32, device="cuda", dtype=torch.int32)
        plans = {"pair4": make_plan(B, "pair4"), "quad32": make_plan(B, "quad32")}
        outs = {}
        for name, plan in plans.items():
            qv.zero_(); rv.zero_(); pm.zero_()
            capi.check(capi.lib().qrk_bd_factorize(plan, tiles.data_ptr(), qv.data_ptr(), rv.data_ptr(), pm.data_ptr(), None, capi.MEM_DEVICE))
            torch.cuda.synchronize()
            outs[name] = (pm[:B * 32].clone(), rv[:B * 528].clone(), qv[:B * 1024].clone())
        same = [bool(torch.equal(x, y)) for x, y in zip(outs["pair4"], outs["quad32"])]
        dq = (outs["pair4"][2] - outs["quad32"][2]).abs().max().item()
        dr = (outs["pair4"][1] - outs["quad32"][1]).abs().max().item()

        def run(plan, it):
            ms = C.c_float()
            capi.check(capi.lib().qrk_bd_time_factorize(plan, tiles.data_ptr(), qv.data_ptr(), rv.data_ptr(), pm.data_ptr(), S, it, C.byref(ms)))
            return ms.value * 1e3
        res = {k: [] for k in ("pair4", "quad32 staged", "quad32 direct")}
        for p in plans.values():
            run(p, 20)
        it = 200 if B <= 20000 else 50
        for _ in range(3):
            res["pair4"].append(run(plans["pair4"], it))
            os.environ["QRK_Q32_DIRECT"] = "0"
            res["quad32 staged"].append(run(plans["quad32"], it))
            os.environ["QRK_Q32_DIRECT"] = "1"
            res["quad32 direct"].append(run(plans["quad32"], it))
            os.environ.pop("QRK_Q32_DIRECT", None)
        line = "  ".join(f"{k} {min(v):8.2f}" for k, v in res.items())
        print(f"B={B:7d}  {line}  us | perm/R/Q bitwise {same}  max|dR| {dr:.2e} max|dQ| {dq:.2e}", flush=True)
        for p in plans.values():
            capi.lib().qrk_bd_plan_destroy(p)


main()
