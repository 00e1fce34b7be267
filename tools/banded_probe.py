"""Timing probe for the block-banded chain at the BASELINE configs[2] shape (cfg 3):
strip i = dense 256 x 192 (4x3 tiles of 64x64) at rows 256 i, columns 64 i; N strips (50 000 in BASELINE).
Usage (GPU box): python tools/banded_probe.py [N]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, scipy.sparse as sp, torch
import qrkit_amd as qa

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(3)
ncols = 64 * N
rows, cols = [], []
for i in range(N):
    w = min(192, ncols - 64 * i)
    r, c = np.meshgrid(np.arange(256 * i, 256 * i + 256), np.arange(64 * i, 64 * i + w), indexing="ij")
    rows.append(r.ravel()); cols.append(c.ravel())
rows = np.concatenate(rows); cols = np.concatenate(cols)
J = sp.csr_matrix((rng.uniform(0.5, 5.0, len(rows)), (rows, cols)), shape=(256 * N, ncols))
J.sort_indices()
qr = qa.BandedBlockedSparseQR(suggestedBlockCols=2)
t0 = time.perf_counter(); qr.analyzePattern(J); t1 = time.perf_counter()
qr.factorize(J); torch.cuda.synchronize()
t2 = time.perf_counter()
reps = 2
for _ in range(reps):
    qr.factorize(J)
torch.cuda.synchronize()
t3 = time.perf_counter()
nb = len(qr.blocks)
print(f"N={N} strips: matrix {J.shape}, nnz {J.nnz}, {nb} merged blocks, first {tuple(qr.blocks[0])}, last {tuple(qr.blocks[-1])}")
print(f"analyzePattern {1e3 * (t1 - t0):.1f} ms (host);  factorize {1e3 * (t3 - t2) / reps:.1f} ms = {1e3 * (t3 - t2) / reps / nb:.3f} ms per panel"
      f"  -> {50000 * (t3 - t2) / reps / nb:.1f} s for 50 000 strips")
x = rng.uniform(-1, 1, ncols)
b = (J @ x)[np.argsort(qr.rowsPermutation())] if qr.hasPermutation else J @ x
xs = qr.solve(b); torch.cuda.synchronize()
t4 = time.perf_counter(); xs = qr.solve(b); torch.cuda.synchronize(); t5 = time.perf_counter()
bt = torch.as_tensor(b, device="cuda")
t6 = time.perf_counter(); qr.applyQt(bt); torch.cuda.synchronize(); t7 = time.perf_counter()
print(f"solve (1 rhs, host vector in/out) {1e3 * (t5 - t4):.1f} ms; Q^T b on the device {1e3 * (t7 - t6):.1f} ms = {1e3 * (t7 - t6) / nb:.3f} ms per panel")
print("LS recovery rel. error", np.linalg.norm(xs - x) / np.linalg.norm(x))
if os.environ.get("QRK_BB_PROF"):
    n = int(qr.blocks[-1][3])
    tv = qr._t[qr._tlen - n * n: qr._tlen - n * n + 14].cpu().numpy()
    names = ["scatter+leftover", "householder QR", "R/leftover/Y out", "Gram Y^T Y", "T recurrence + write", "  (of QR: step heads incl. x' build)", "  QR A load sub-panel", "  QR B sub-panel QR", "  QR C+D store, larft", "  QR E (barrier waits)", "   E pass 1 w = V^T W", "   E reduce over row parts", "   E u = -T^T w", "   E pass 2 W += V u"]
    print("per panel (us at 100 MHz s_memtime? ticks / panels):", {k: round(float(v) / nb, 1) for k, v in zip(names, tv)})
