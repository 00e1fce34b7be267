"""Time of one factorisation of B tiles of +-1 entries (every tile takes the in-kernel exact path) through both generations of the 32 x 32 kernel."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import p4_check as pc
import qrkit_amd
ctx = qrkit_amd.Context(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
g = torch.Generator(device="cuda").manual_seed(1)
for frac in (1.0, 0.1, 0.01):
    t = torch.rand(B, 1024, device="cuda", dtype=torch.float64, generator=g) * 4.5 + 0.5
    n = int(B * frac)
    t[:n] = torch.randint(0, 2, (n, 1024), device="cuda", generator=g).double() * 2 - 1
    t = t[torch.randperm(B, device="cuda", generator=g)].reshape(-1).contiguous()
    for v2 in (False, True):
        plan = pc.make_plan(ctx, B, True, v2)
        pc.factor(plan, B, t)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            pc.factor(plan, B, t)
        dt = (time.perf_counter() - t0) / 3
        print(f"B={B} tie fraction {frac:5.2f}  {'gen2' if v2 else 'gen1'}: {dt * 1e3:8.3f} ms per factorisation", flush=True)
