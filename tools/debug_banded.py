import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
import qrkit_amd
from oracle import oracle as orc
from test_banded import banded_matrix
J = banded_matrix(64, True, None)
ref = orc.bb_factorize(J, 8)
qr = qrkit_amd.BandedBlockedSparseQR(suggestedBlockCols=8); qr.compute(J)
print("blocks", qr.blocks[:3].tolist(), "yty", qr.yty[:3].tolist())
for k in (0, 1):
    Y, T, row, nz = qr.blockYTY(k); Yo, To, _, _ = ref.yty[k]
    E = np.abs(Y - Yo)
    print("block", k, "Y shape", Y.shape, "max err", E.max(), "at", np.unravel_index(E.argmax(), E.shape))
    print(" rows with err:", np.where(E.max(axis=1) > 1e-10)[0], "cols:", np.where(E.max(axis=0) > 1e-10)[0])
    print(" T err", np.abs(T - To).max())
import scipy.linalg as sl
pm = J.toarray()
Ji = pm[0:21, 0:8].copy()
qro, hco = orc.householder_qr(Ji)
(qrs, taus), _ = sl.qr(Ji, mode='raw')
print("oracle vs lapack packed:", np.abs(qro - qrs).max(), "tau", np.abs(hco - taus).max())
Y, T, row, nz = qr.blockYTY(0)
print("product Y col7 rows 8..12:", Y[8:13, 7], "\noracle  :", qro[8:13, 7], "\nlapack  :", qrs[8:13, 7])
print("product T diag:", np.diag(T), "\noracle hc:", hco)
