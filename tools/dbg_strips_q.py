import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import qrkit_amd
from qrkit_amd import _capi as capi
from qrkit_amd.banded import BandedStripsQR
N, ms, n, s = int(sys.argv[1]), 64, 48, 16
rng = np.random.default_rng(1)
strips = rng.uniform(-1, 1, (N, ms, n))
dev = torch.from_numpy(np.ascontiguousarray(strips.transpose(0, 2, 1)).reshape(-1)).cuda()
qr = BandedStripsQR(N, ms, n, s, context=qrkit_amd.Context(0))
qr.factorize(dev)
rows = qr.rows()
b = torch.from_numpy(rng.uniform(-1, 1, rows)).cuda()
def app(v, tr, poison):
    out = torch.full((rows,), float("nan"), dtype=torch.float64, device="cuda")
    work = torch.full((rows,), float("nan") if poison else 0.0, dtype=torch.float64, device="cuda")
    src = v.clone()
    qr._ctx.use_current_stream()
    capi.check(capi.lib().qrk_bbs_apply_q(qr._plan, 1 if tr else 0, src.data_ptr(), out.data_ptr(), 1, work.data_ptr()), qr._ctx.handle)
    torch.cuda.synchronize()
    return out, work
for sw in ("0", "1"):
    os.environ["QRK_BBS_MAPS"] = sw
    y, w1 = app(b, True, True)
    z, w2 = app(y, False, True)
    print(sw, "nan in y", int(torch.isnan(y).sum()), "nan in z", int(torch.isnan(z).sum()), "nan in work(Q dir)", int(torch.isnan(w2).sum()),
          "err", float((z - b).norm() / b.norm()))
    if int(torch.isnan(w2).sum()):
        idx = torch.nonzero(torch.isnan(w2)).flatten().cpu().numpy()
        print("   first nan positions in work:", idx[:20], "strip", idx[:20] // ms, "row in strip", idx[:20] % ms)
