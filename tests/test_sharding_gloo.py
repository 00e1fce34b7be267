"""world_size-2 test of the multi-GPU path on CPU (gloo): shard -> per-rank factorisation -> gather of R / perm
reproduces the unsharded factorisation.  The per-rank worker here is an oracle-backed stand-in with the solver
interface (this is a test of the sharding and gather logic, not of the kernels)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _OracleSolver:
    def compute(self, mat):
        from oracle import oracle as orc
        self._res = orc.BDProblem(mat["rows"], mat["cols"], mat["tiles"]).factorize()

    def rValues(self):
        return torch.from_numpy(self._res.R_vals.copy())

    def qValues(self):
        return torch.from_numpy(self._res.Q_vals.copy())

    def colsPermutation(self):
        return self._res.perm


def _worker(rank, world, port, rows, cols, tiles, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qrkit_amd.sharding import ShardedBlockDiagonalQR
    sh = ShardedBlockDiagonalQR(rows, cols, rank, world, solver_factory=_OracleSolver)
    lr, lc = sh.local_layout()
    sizes = rows.astype(np.int64) * cols
    t0, t1 = int(sizes[:sh.start].sum()), int(sizes[:sh.end].sum())
    sh.compute({"rows": lr, "cols": lc, "tiles": tiles[t0:t1]})
    R = sh.gatherR().numpy()
    P = sh.gatherPerm().numpy()
    if rank == 0:
        np.save(out + "_R.npy", R); np.save(out + "_P.npy", P)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2(tmp_path):
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    rng = np.random.default_rng(2)
    B = 37
    cols = rng.integers(1, 20, B).astype(np.int32)
    rows = (cols + rng.integers(0, 5, B)).astype(np.int32)
    tiles = orc.gen_uniform(9, -1.0, 1.0, int((rows.astype(np.int64) * cols).sum()))
    ref = orc.BDProblem(rows, cols, tiles).factorize()
    out = str(tmp_path / "g")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, rows, cols, tiles, out), nprocs=2, join=True)
    np.testing.assert_array_equal(np.load(out + "_P.npy"), ref.perm)
    np.testing.assert_array_equal(np.load(out + "_R.npy"), ref.R_vals)
