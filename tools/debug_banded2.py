import sys, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import qrkit_amd
from oracle import oracle as orc
from test_banded import banded_matrix
J = banded_matrix(256, True, 3)
ref = orc.bb_factorize(J, 8)
qr = qrkit_amd.BandedBlockedSparseQR(suggestedBlockCols=8); qr.compute(J)
R = qr.matrixR().toarray(); Ro = ref.R.toarray()
E = np.abs(R - Ro)
rows = np.where(E.max(axis=1) > 1e-9 * np.abs(Ro).max())[0]
print("rows with error:", rows[:20], "count", len(rows), "of", R.shape)
print("blocks:", qr.blocks[:6].tolist())
if len(rows):
    r0 = rows[0]
    cols = np.where(E[r0] > 1e-9)[0]
    print("first bad row", r0, "cols", cols[:10], "got", R[r0, cols[:6]], "want", Ro[r0, cols[:6]])
for k in range(min(6, len(qr.blocks))):
    Y, T, row, nz = qr.blockYTY(k); Yo, To, _, _ = ref.yty[k]
    print("block", k, qr.blocks[k].tolist(), "Y err", np.abs(Y - Yo).max(), "T err", np.abs(T - To).max(), "Y shape", Y.shape)
