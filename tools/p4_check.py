"""K1 second generation (bdqr_pair4.hip, QRK_PAIR_V2=1) against the shipped K1 in one process: parity (perm bit-exact, Q / R to 1e-12 of
each other and orthogonality / reconstruction checked on their own), then time per launch of both at the headline size.

  python tools/p4_check.py [B]
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import qrkit_amd                                   # noqa: E402
from qrkit_amd import _capi as capi                # noqa: E402


def make_plan(ctx, B, pivoting, v2):
    os.environ["QRK_PAIR_V2"] = "1" if v2 else "0"
    lay = capi.BDLayout()
    lay.num_blocks, lay.block_rows, lay.block_cols = B, 32, 32
    lay.rows = lay.cols = None
    lay.mat_rows = lay.mat_cols = B * 32
    plan = C.c_void_p()
    capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), 0, 0 if pivoting else 1, C.byref(plan)))
    os.environ.pop("QRK_PAIR_V2", None)
    return plan


def factor(plan, B, tiles):
    qv = torch.full((B * 1024,), float("nan"), device="cuda", dtype=torch.float64)
    rv = torch.full((B * 528,), float("nan"), device="cuda", dtype=torch.float64)
    pm = torch.full((B * 32,), -1, device="cuda", dtype=torch.int32)
    capi.check(capi.lib().qrk_bd_factorize(plan, tiles.data_ptr(), qv.data_ptr(), rv.data_ptr(), pm.data_ptr(), None, capi.MEM_DEVICE))
    torch.cuda.synchronize()
    return qv, rv, pm


def check(ctx, B, pivoting, tiles, label):
    p1, p2 = make_plan(ctx, B, pivoting, False), make_plan(ctx, B, pivoting, True)
    q1, r1, m1 = factor(p1, B, tiles)
    q2, r2, m2 = factor(p2, B, tiles)
    same_perm = bool((m1 == m2).all())
    dq = (q1 - q2).abs().max().item()
    dr = (r1 - r2).abs().max().item()
    # on its own: Q^T Q = I, Q R = A P
    Q = q2.view(B, 32, 32)
    A = tiles.view(B, 32, 32).transpose(1, 2)                   # tiles are column-major
    R = torch.zeros(B, 32, 32, device="cuda", dtype=torch.float64)
    iu = torch.triu_indices(32, 32, device="cuda")
    order = torch.argsort(iu[1] * 32 + iu[0])                   # packed CSC order: column by column
    R[:, iu[0][order], iu[1][order]] = r2.view(B, 528)
    P = (m2.view(B, 32) - (torch.arange(B, device="cuda", dtype=torch.int32) * 32)[:, None]).long()
    AP = torch.gather(A, 2, P[:, None, :].expand(B, 32, 32))
    orth = (Q.transpose(1, 2) @ Q - torch.eye(32, device="cuda", dtype=torch.float64)).abs().max().item()
    rec = (Q @ R - AP).abs().max().item() / A.abs().max().item()
    print(f"{label:28s} B={B:6d} piv={int(pivoting)} perm_equal={same_perm} |dQ|={dq:.2e} |dR|={dr:.2e} orth={orth:.2e} rec={rec:.2e}", flush=True)
    lib = capi.lib()
    lib.qrk_bd_plan_destroy(p1); lib.qrk_bd_plan_destroy(p2)
    return same_perm and dq < 1e-11 and dr < 1e-10 and orth < 1e-13 and rec < 1e-13


def timeit(ctx, B, v2, pivoting=True):
    plan = make_plan(ctx, B, pivoting, v2)
    S = max(1, min(8, (80000 + B - 1) // B))
    g = torch.Generator(device="cuda").manual_seed(1)
    tiles = torch.rand(S * B * 1024, device="cuda", dtype=torch.float64, generator=g) * 4.5 + 0.5
    qv = torch.empty(S * B * 1024, device="cuda", dtype=torch.float64)
    rv = torch.empty(S * B * 528, device="cuda", dtype=torch.float64)
    pm = torch.empty(S * B * 32, device="cuda", dtype=torch.int32)
    ms = C.c_float()

    def run(it):
        capi.check(capi.lib().qrk_bd_time_factorize(plan, tiles.data_ptr(), qv.data_ptr(), rv.data_ptr(), pm.data_ptr(), S, it, C.byref(ms)))
        return ms.value * 1e3
    run(30)
    v = [run(200) for _ in range(3)]
    capi.lib().qrk_bd_plan_destroy(plan)
    return min(v)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    ctx = qrkit_amd.Context(0)
    ok = True
    g = torch.Generator(device="cuda").manual_seed(7)
    for n in (1, 2, 3, 7, 513, B):
        t = torch.rand(n * 1024, device="cuda", dtype=torch.float64, generator=g) * 4.5 + 0.5
        ok &= check(ctx, n, True, t, "uniform(0.5,5)")
        ok &= check(ctx, n, False, t, "uniform(0.5,5) no pivoting")
    t = torch.randn(2000 * 1024, device="cuda", dtype=torch.float64, generator=g)
    ok &= check(ctx, 2000, True, t, "normal")
    # exact ties / rank deficiency: the flagged tiles take the kernel's own exact path
    t = torch.randint(-2, 3, (600 * 1024,), device="cuda", generator=g).double()
    ok &= check(ctx, 600, True, t, "small integers (ties)")
    t = torch.rand(300, 32, 32, device="cuda", dtype=torch.float64, generator=g)
    t[:, 5] = t[:, 9]; t[:, 17] = 0.0; t[100:, 20:] = 0.0
    ok &= check(ctx, 300, True, t.reshape(-1).contiguous(), "duplicate / zero columns")
    ok &= check(ctx, 300, False, t.reshape(-1).contiguous(), "duplicate / zero columns np")
    print("PARITY", "OK" if ok else "FAILED", flush=True)
    for b in (B, 100000):
        for piv in (True, False):
            a = [timeit(ctx, b, False, piv), timeit(ctx, b, True, piv), timeit(ctx, b, False, piv), timeit(ctx, b, True, piv)]
            print(f"B={b} piv={int(piv)}: K1 {a[0]:.2f} {a[2]:.2f} us   v2 {a[1]:.2f} {a[3]:.2f} us", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
