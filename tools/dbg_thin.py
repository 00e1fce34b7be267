import sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, __import__("os").environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, __import__("os").environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tests")
import qrkit_amd
from oracle import oracle as orc
from test_thin_gpu import thin_sparse_problem
rows, cols, bc, seed = 150, 24, 2, 5
M = sp.lil_matrix(thin_sparse_problem(rows, cols, seed)); M[:, 5] = M[:, 3]; M[:, 11] = 0.0
M = sp.csc_matrix(M); M.eliminate_zeros()
ref = orc.bt_sparse_qr(M, bc)
qr = qrkit_amd.BlockedThinSparseQR(qrkit_amd.Context(0), bc); qr.compute(M)
R = qr.matrixR().cpu().numpy()
D = np.abs(R - ref.R)
print("rank", qr.rank(), ref.rank, "max diff", D.max(), "at", np.unravel_index(D.argmax(), D.shape), "normR", np.linalg.norm(ref.R))
bad = np.argwhere(D > 1e-10)
print(bad[:20])
for (i, j) in bad[:8]:
    print(i, j, R[i, j], ref.R[i, j])
np.set_printoptions(precision=4, linewidth=250, suppress=True)
print("perm", qr.colsPermutation().cpu().numpy())
for i in range(11, 19):
    print("mine", i, R[i, 10:24]); print("ref ", i, ref.R[i, 10:24])
from test_thin_gpu import permuted
PM = permuted(M, ref.perm, ref.rowperm)
QtPM = qr._applyAny(PM, True)
print("QtPM-R", np.linalg.norm(QtPM[:cols] - R[:cols]), "below", np.linalg.norm(QtPM[cols:]), "normPM", np.linalg.norm(PM))
sv = np.linalg.svd(M.toarray(), compute_uv=False); svr = np.linalg.svd(R[:cols], compute_uv=False)
print("sv diff", np.abs(sv - svr).max(), sv[:3], svr[:3], sv[-3:], svr[-3:])
print("ref: QtPM-R", np.linalg.norm(orc.bt_apply_q(ref, PM, True)[:cols] - ref.R[:cols]))
