"""Multi-process tests of the multi-GPU path (gloo): shard -> per-rank factorisation -> gather of R / perm reproduces the
unsharded factorisation; the sharded solve (block-local, x only gathered) reproduces the un-sharded _solve_impl; the overlapped gather
(pieces of the range, a piece's R on its way while the next piece is factorised) reproduces the plain one.

* CPU (world 2 and 3): the per-rank worker is an oracle-backed stand-in with the solver interface -- a test of the sharding and
  of the ragged gathers (to every rank, and to the root only with true byte counts), not of the kernels.
* GPU (`-m gpu`, world 2 on the one card of the box): the SAME code with the real HIP solver under ShardedBlockDiagonalQR,
  collectives over gloo (two ranks cannot share one device in an RCCL communicator); compared with the oracle.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _OracleSolver:
    def compute(self, mat):
        from oracle import oracle as orc
        self._prob = orc.BDProblem(mat["rows"], mat["cols"], mat["tiles"])
        self._res = self._prob.factorize()

    def solve(self, b):
        return torch.from_numpy(np.ascontiguousarray(self._prob.solve(self._res, np.asarray(b, dtype=np.float64))))

    def rValues(self):
        return torch.from_numpy(self._res.R_vals.copy())

    def qValues(self):
        return torch.from_numpy(self._res.Q_vals.copy())

    def colsPermutation(self):
        return self._res.perm


def _worker(rank, world, port, rows, cols, tiles, b, out, use_gpu):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qrkit_amd.sharding import ShardedBlockDiagonalQR
    sizes = rows.astype(np.int64) * cols
    if use_gpu:
        import qrkit_amd
        sh = ShardedBlockDiagonalQR(rows, cols, rank, world)          # default factory: the HIP solver on cuda:0
        lr, lc = sh.local_layout()
        t0, t1 = int(sizes[:sh.start].sum()), int(sizes[:sh.end].sum())
        sh.compute(qrkit_amd.SparseBlockDiagonal.fromTiles(lr, lc, tiles[t0:t1]))
        cpu = lambda t: None if t is None else t.cpu()
        # gloo moves host tensors: stage the device shards through the host for this rehearsal
        class _Host:
            def __init__(s, inner): s.i = inner
            def rValues(s): return s.i.rValues().cpu()
            def colsPermutation(s): return s.i.colsPermutation()
        sh.solver = _Host(sh.solver)
    else:
        sh = ShardedBlockDiagonalQR(rows, cols, rank, world, solver_factory=_OracleSolver)
        lr, lc = sh.local_layout()
        t0, t1 = int(sizes[:sh.start].sum()), int(sizes[:sh.end].sum())
        sh.compute({"rows": lr, "cols": lc, "tiles": tiles[t0:t1]})
    R = sh.gatherR().numpy()                 # on every rank
    P = sh.gatherPerm().numpy()
    R0 = sh.gatherR(root=0)                  # on the root only, true byte counts
    P0 = sh.gatherPerm(root=0)
    assert (R0 is None) == (rank != 0) and (P0 is None) == (rank != 0)
    np.save(f"{out}_R{rank}.npy", R); np.save(f"{out}_P{rank}.npy", P)
    if rank == 0:
        np.save(out + "_R0.root.npy", R0.numpy()); np.save(out + "_P0.root.npy", P0.numpy())
    # the sharded solve: block-local solve, x only gathered (every rank / the root only; one and three right-hand sides)
    r0, r1 = int(rows[:sh.start].sum()), int(rows[:sh.end].sum())
    if use_gpu:
        sh.solver = sh.solver.i
        class _HostX:
            def __init__(s, inner): s.i = inner
            def rValues(s): return s.i.rValues().cpu()
            def colsPermutation(s): return s.i.colsPermutation()
            def solve(s, b): return torch.as_tensor(s.i.solve(b)).cpu()
        sh.solver = _HostX(sh.solver)
    x_all = sh.solve(torch.from_numpy(b[r0:r1]))
    x_root = sh.solve(torch.from_numpy(b[r0:r1]), root=0)
    x3 = sh.solve(torch.from_numpy(np.stack([b[r0:r1], 2.0 * b[r0:r1], -b[r0:r1]], axis=1)))
    assert (x_root is None) == (rank != 0)
    np.save(f"{out}_x{rank}.npy", x_all.numpy()); np.save(f"{out}_x3_{rank}.npy", x3.numpy())
    if rank == 0:
        np.save(out + "_x.root.npy", x_root.numpy())
    # the overlapped gather: the range in three pieces, a piece's R on its way while the next one is factorised; then the solve
    # through the pieces
    if use_gpu:
        import qrkit_amd
        class _HostSolver(qrkit_amd.BlockDiagonalSparseQR):
            def rValues(s): return super().rValues().cpu()
            def solve(s, bb): return torch.as_tensor(super().solve(bb)).cpu()
        local = qrkit_amd.SparseBlockDiagonal.fromTiles(lr, lc, tiles[t0:t1])
        Rov = sh.computeGatherR(local, chunks=3, root=0, solver_factory=_HostSolver)
    else:
        cut = lambda m, a, e: {"rows": m["rows"][a:e], "cols": m["cols"][a:e],
                               "tiles": m["tiles"][int((m["rows"][:a].astype(np.int64) * m["cols"][:a]).sum()):
                                                   int((m["rows"][:e].astype(np.int64) * m["cols"][:e]).sum())]}
        Rov = sh.computeGatherR({"rows": lr, "cols": lc, "tiles": tiles[t0:t1]}, chunks=3, root=0, slice_mat=cut, solver_factory=_OracleSolver)
    assert (Rov is None) == (rank != 0)
    if rank == 0:
        np.save(out + "_Rov.npy", Rov.numpy())
    np.save(f"{out}_Pov{rank}.npy", sh.gatherPerm().numpy())
    np.save(f"{out}_xov{rank}.npy", sh.solve(torch.from_numpy(b[r0:r1])).numpy())
    dist.barrier()
    dist.destroy_process_group()


def _run(tmp_path, world, use_gpu, seed=2, B=37):
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    rng = np.random.default_rng(seed)
    cols = rng.integers(1, 20, B).astype(np.int32)
    rows = (cols + rng.integers(0, 5, B)).astype(np.int32)
    tiles = orc.gen_uniform(9, -1.0, 1.0, int((rows.astype(np.int64) * cols).sum()))
    prob = orc.BDProblem(rows, cols, tiles)
    ref = prob.factorize()
    b = rng.uniform(-1.0, 1.0, int(rows.sum()))
    xref = prob.solve(ref, b)
    out = str(tmp_path / "g")
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, rows, cols, tiles, b, out, use_gpu), nprocs=world, join=True)
    for rank in range(world):
        np.testing.assert_array_equal(np.load(f"{out}_P{rank}.npy"), ref.perm)
        if use_gpu:
            assert np.linalg.norm(np.load(f"{out}_R{rank}.npy") - ref.R_vals) <= 1e-12 * np.linalg.norm(ref.R_vals)
        else:
            np.testing.assert_array_equal(np.load(f"{out}_R{rank}.npy"), ref.R_vals)
    np.testing.assert_array_equal(np.load(out + "_P0.root.npy"), ref.perm)
    np.testing.assert_array_equal(np.load(out + "_R0.root.npy"), np.load(f"{out}_R0.npy"))
    # the sharded solve (x only gathered) = the un-sharded _solve_impl; the overlapped gather = the plain one
    tol = 1e-9 if use_gpu else 0.0
    for rank in range(world):
        for name, want in (("x", xref), ("xov", xref)):
            got = np.load(f"{out}_{name}{rank}.npy")
            assert got.shape == xref.shape and np.linalg.norm(got - want) <= tol * np.linalg.norm(want), (name, rank)
        x3 = np.load(f"{out}_x3_{rank}.npy")
        assert x3.shape == (xref.size, 3)
        for k, f in enumerate((1.0, 2.0, -1.0)):
            assert np.linalg.norm(x3[:, k] - f * xref) <= max(tol, 1e-13) * np.linalg.norm(xref)
        np.testing.assert_array_equal(np.load(f"{out}_Pov{rank}.npy"), ref.perm)
    np.testing.assert_array_equal(np.load(out + "_x.root.npy"), np.load(f"{out}_x0.npy"))
    if use_gpu:
        assert np.linalg.norm(np.load(out + "_Rov.npy") - ref.R_vals) <= 1e-12 * np.linalg.norm(ref.R_vals)
    else:
        np.testing.assert_array_equal(np.load(out + "_Rov.npy"), ref.R_vals)


@pytest.mark.parametrize("world", [2, 3])
def test_shard_and_gather(tmp_path, world):
    _run(tmp_path, world, use_gpu=False)


@pytest.mark.gpu
def test_shard_and_gather_world2_hip_solver(tmp_path):
    """The HIP solver under ShardedBlockDiagonalQR, two ranks on the box's one GPU, gloo for the exchange."""
    _run(tmp_path, 2, use_gpu=True, seed=5, B=61)


def _angular_worker(rank, world, port, rows, cols, tiles, J2, b, extra, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import qrkit_amd
    from qrkit_amd.sharding import ShardedBlockAngularQR
    m2 = J2.shape[1]
    sh = ShardedBlockAngularQR(rows, cols, m2, rank, world)
    lr, lc = sh.local_layout()
    sizes = rows.astype(np.int64) * cols
    t0, t1 = int(sizes[:sh.start].sum()), int(sizes[:sh.end].sum())
    r0, r1 = int(rows[:sh.start].sum()), int(rows[:sh.end].sum())
    n1 = int(rows.sum())
    rsel = np.arange(r0, r1) if rank < world - 1 else np.concatenate([np.arange(r0, r1), np.arange(n1, n1 + extra)])
    left = qrkit_amd.SparseBlockDiagonal.fromTiles(lr, lc, tiles[t0:t1])
    sh.compute(left, torch.from_numpy(J2[rsel, :]))
    x1, x2 = sh.solve(torch.from_numpy(b[rsel]))
    np.save(f"{out}_x1_{rank}.npy", x1.cpu().numpy())
    np.save(f"{out}_x2_{rank}.npy", x2.cpu().numpy())
    np.save(f"{out}_p2_{rank}.npy", sh.colsPermutationRight())
    np.save(f"{out}_red_{rank}.npy", np.array([int(sh._reduced)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("B,m2,extra,reduced,world", [(600, 160, 37, True, 2), (40, 64, 5, False, 2), (900, 96, 0, True, 3)])
def test_sharded_block_angular_tsqr_world2(tmp_path, B, m2, extra, reduced, world):
    """ShardedBlockAngularQR, two or three ranks on the box's one GPU (gloo): left tiles and the rows of J2 sharded, the bottom rows reduced
    per rank to a triangle (qrk_tsqr_*), the triangles stacked and pivoted on the root; the least-squares solution and the
    right-block permutation against the un-sharded BlockAngularSparseQR (BlockAngularSparseQR.h:459-514, :202-227) run by the
    oracle.  Second shape: ranks with fewer bottom rows than right columns send the rows themselves."""
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    rng = np.random.default_rng(B + m2)
    r, c = 8, 6
    rows = np.full(B, r, np.int32); cols = np.full(B, c, np.int32)
    tiles = rng.uniform(0.5, 5.0, B * r * c)
    n1, m1 = B * r, B * c
    J2 = rng.uniform(0.5, 5.0, (n1 + extra, m2))
    x = rng.uniform(-1.0, 1.0, m1 + m2)
    T = tiles.reshape(B, c, r)
    b = J2 @ x[m1:]
    b[:n1] += np.einsum("bcr,bc->br", T, x[:m1].reshape(B, c)).reshape(-1)
    out = str(tmp_path / "a")
    port = 31500 + (os.getpid() % 2000) + B % 7
    mp.spawn(_angular_worker, args=(world, port, rows, cols, tiles, J2, b, extra, out), nprocs=world, join=True)
    x1 = np.concatenate([np.load(f"{out}_x1_{k}.npy") for k in range(world)])
    x2 = np.load(f"{out}_x2_0.npy")
    for k in range(1, world):
        np.testing.assert_array_equal(x2, np.load(f"{out}_x2_{k}.npy"))
    xs = np.concatenate([x1, x2])
    assert np.linalg.norm(xs - x) <= 1e-10 * np.linalg.norm(x)
    assert bool(np.load(f"{out}_red_0.npy")[0]) == reduced
    # the right-block permutation is the un-sharded one: the oracle's BlockAngularSparseQR on the whole matrix
    ref = orc.ba_factorize(orc.BDProblem(rows, cols, tiles), J2)
    np.testing.assert_array_equal(np.load(f"{out}_p2_0.npy") + m1, ref.perm[m1:])
    for k in range(1, world):
        np.testing.assert_array_equal(np.load(f"{out}_p2_0.npy"), np.load(f"{out}_p2_{k}.npy"))
