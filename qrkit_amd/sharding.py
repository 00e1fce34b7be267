"""Block sharding across the GPUs of one node (one process per GPU, torch.distributed / RCCL).

The diagonal blocks of a SparseBlockDiagonal are independent (the hot loop of
BlockDiagonalSparseQR::factorize, BlockDiagonalSparseQR.h:432, carries only the running offsets
base_row/base_col), so they shard as contiguous ranges with NO data-path collective: every rank
factorises its range and its Q / R / perm shards are already in final global order.  The only
exchange is the optional gather of the composed R (and perm) shards for a caller that needs the
whole factor (grouped send/recv with true byte counts over RCCL/xGMI -- to the root only, or to every rank; `gloo` in the
CPU tests).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_ranges(block_rows: Sequence[int], block_cols: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Contiguous block ranges [start, end) per rank, balanced by the Householder cost r*c^2 per block
    (equal counts for uniform blocks).  Ranges are contiguous so that output offsets stay prefix sums."""
    r = np.asarray(block_rows, dtype=np.float64)
    c = np.asarray(block_cols, dtype=np.float64)
    B = len(r)
    if B == 0:
        return [(0, 0)] * world
    cost = np.cumsum(r * c * c)
    total = cost[-1]
    bounds = [0]
    for k in range(1, world):
        target = total * k / world
        idx = int(np.searchsorted(cost, target, side="left")) + 1
        # pick the cut closest to the target
        if idx - 1 > bounds[-1] and abs(cost[idx - 2] - target) <= abs(cost[idx - 1] - target):
            idx -= 1
        bounds.append(min(max(idx, bounds[-1]), B))
    bounds.append(B)
    return [(bounds[k], bounds[k + 1]) for k in range(world)]


def shard_offsets(block_rows, block_cols, start: int, end: int):
    """Global offsets of a shard: (base_row, base_col, q_offset, r_offset) of its first block."""
    r = np.asarray(block_rows[:start], dtype=np.int64)
    c = np.asarray(block_cols[:start], dtype=np.int64)
    return int(r.sum()), int(c.sum()), int((r * r).sum()), int((c * (c + 1) // 2).sum())


def exchange_sizes(n_local: int, device, group=None) -> List[int]:
    """Element counts of every rank's shard (one small all_gather of an int64)."""
    world = dist.get_world_size(group)
    n = torch.tensor([n_local], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    return [int(s.item()) for s in sizes]


def gather_ragged_to_root(local: torch.Tensor, sizes: Sequence[int], out, root: int, rank: int, world: int, group=None):
    """Rank-ordered concatenation of 1-D shards of different lengths ON THE ROOT ONLY, with true byte counts: every peer posts
    one send of exactly its shard, the root one receive per peer straight into its slice of `out` (one grouped launch:
    ncclGroupStart/End + ncclSend/ncclRecv over RCCL, so all links into the root are busy at once; isend/irecv over gloo).
    No padding to the largest shard and nothing travels to ranks that do not need it."""
    if world == 1:
        out[:local.numel()].copy_(local)
        return out
    ops = []
    if rank == root:
        off = 0
        for peer in range(world):
            n = int(sizes[peer])
            if peer == root:
                out[off:off + n].copy_(local)
            elif n > 0:
                ops.append(dist.P2POp(dist.irecv, out[off:off + n], peer, group))
            off += n
    elif local.numel() > 0:
        ops.append(dist.P2POp(dist.isend, local.contiguous(), root, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out


def all_gather_ragged(local: torch.Tensor, group=None, sizes: Sequence[int] = None) -> torch.Tensor:
    """Concatenate 1-D shards of different lengths from all ranks, in rank order, on EVERY rank: each shard crosses one link
    to each peer with its true byte count (grouped send/recv), instead of an all_gather padded to the largest shard."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if sizes is None:
        sizes = exchange_sizes(local.numel(), local.device, group)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    out = torch.empty(int(offs[-1]), dtype=local.dtype, device=local.device)
    out[offs[rank]:offs[rank + 1]].copy_(local)
    ops = []
    src = local.contiguous()
    for peer in range(world):
        if peer == rank:
            continue
        if sizes[peer] > 0:
            ops.append(dist.P2POp(dist.irecv, out[offs[peer]:offs[peer + 1]], peer, group))
        if sizes[rank] > 0:
            ops.append(dist.P2POp(dist.isend, src, peer, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out


class ShardedBlockDiagonalQR:
    """Each rank factorises its contiguous block range; gatherR()/gatherPerm() compose the global factor.

    `solver_factory()` returns an object with the BlockDiagonalSparseQR interface (compute, rValues,
    qValues, colsPermutation); the default is the HIP solver on this rank's GPU.
    """

    def __init__(self, block_rows, block_cols, rank: int, world: int, solver_factory=None, group=None):
        self.block_rows = np.asarray(block_rows, dtype=np.int32)
        self.block_cols = np.asarray(block_cols, dtype=np.int32)
        self.rank_id, self.world, self.group = rank, world, group
        self.ranges = shard_ranges(self.block_rows, self.block_cols, world)
        self.start, self.end = self.ranges[rank]
        self.base_row, self.base_col, self.q_off, self.r_off = shard_offsets(self.block_rows, self.block_cols,
                                                                            self.start, self.end)
        if solver_factory is None:
            from .solvers import BlockDiagonalSparseQR
            solver_factory = BlockDiagonalSparseQR
        self.solver = solver_factory()

    def local_layout(self):
        return self.block_rows[self.start:self.end], self.block_cols[self.start:self.end]

    def compute(self, local_mat):
        self.solver.compute(local_mat)
        return self

    def _r_sizes(self):
        c = self.block_cols.astype(np.int64)
        return [int((c[a:b] * (c[a:b] + 1) // 2).sum()) for a, b in self.ranges]

    def _c_sizes(self):
        return [int(self.block_cols[a:b].astype(np.int64).sum()) for a, b in self.ranges]

    def gatherR(self, root=None) -> torch.Tensor:
        """The composed packed R: on every rank (root=None), or on `root` only (other ranks get None).  Sizes come from the
        block map every rank already holds, so no size exchange precedes the data."""
        local = self.solver.rValues().contiguous()
        if root is None:
            return all_gather_ragged(local, self.group, self._r_sizes())
        sizes = self._r_sizes()
        out = torch.empty(sum(sizes), dtype=local.dtype, device=local.device) if self.rank_id == root else None
        gather_ragged_to_root(local, sizes, out, root, self.rank_id, self.world, self.group)
        return out

    def gatherPerm(self, root=None) -> torch.Tensor:
        """Global m_outputPerm_c indices: shard-local indices shifted by the shard's base_col, on the device (no host round trip
        when the solver exposes the device array)."""
        if hasattr(self.solver, "colsPermutationDevice"):
            p = self.solver.colsPermutationDevice().to(torch.int32)
        else:
            p = torch.as_tensor(self.solver.colsPermutation()).to(self.solver.rValues().device).to(torch.int32)
        p = (p + self.base_col).contiguous()
        if root is None:
            return all_gather_ragged(p, self.group, self._c_sizes())
        sizes = self._c_sizes()
        out = torch.empty(sum(sizes), dtype=torch.int32, device=p.device) if self.rank_id == root else None
        gather_ragged_to_root(p, sizes, out, root, self.rank_id, self.world, self.group)
        return out


class ShardedBlockAngularQR:
    """QRKit::BlockAngularSparseQR with the rows sharded over the GPUs of a node (BASELINE configs[3], "8 x MI355X sharded").

    Rank g holds the tiles [start, end) of the block-diagonal left block J1 and the rows of the dense right block J2 that
    belong to them (the last rank also any rows of J2 below J1).  factorize (BlockAngularSparseQR.h:459-514):
      * J1_g = Q1_g R1_g and T = Q1_g^T J2_g on the rank (no exchange: :472-475, solveRightBlock :361-369);
      * the strip S_g = T(0:m1_g, :) stays on the rank (it is only needed for z1 = R1^-1 (y1 - S z2));
      * the bottom rows of T are reduced on the rank to ONE n x n triangle, bottom_g = Q0_g R0_g (qrk_tsqr_*: un-pivoted CAQR on
        the matrix cores) -- tall-skinny QR across ranks: the root gathers world triangles (n^2 doubles each, against
        rows_g * n for the rows themselves), stacks them and runs the pivoted right solver on the stack (the Gram structure of
        the columns, hence Eigen's pivots, is that of the un-sharded bottom block);
      * the permutation P2 of the right block goes back to every rank.
    solve (_solve_impl :202-227) follows the same route with one n-vector per rank up and z2 down.
    Collectives: torch.distributed on `group` (RCCL as "nccl"; with "gloo" -- the CPU/one-GPU rehearsal -- tensors are staged
    through the host).  No reference site exists for the sharding itself."""

    def __init__(self, block_rows, block_cols, m2: int, rank: int, world: int, context=None, group=None, root: int = 0):
        from .angular import DenseColPivQR, DenseTSQR
        from .solvers import BlockDiagonalSparseQR, Context
        from . import _capi as capi
        self.block_rows = np.asarray(block_rows, dtype=np.int32)
        self.block_cols = np.asarray(block_cols, dtype=np.int32)
        self.m2, self.rank_id, self.world, self.group, self.root = int(m2), rank, world, group, root
        self.ranges = shard_ranges(self.block_rows, self.block_cols, world)
        self.start, self.end = self.ranges[rank]
        if context is None:
            # one process per GPU: the rank's own device (LOCAL_RANK under torch.distributed.run), never "device 0 on every rank"
            # (RCCL rejects two ranks on one device; gloo would silently serialise them)
            import os
            ndev = max(torch.cuda.device_count(), 1)
            context = Context(int(os.environ.get("LOCAL_RANK", rank)) % ndev)
        self._ctx = context
        self.m_leftSolver = BlockDiagonalSparseQR(blockSolver=capi.COLPIV_HOUSEHOLDER, qFormat=capi.FULL_Q, context=self._ctx)
        self._tsqr = DenseTSQR(self._ctx)
        self.m_rightSolver = DenseColPivQR(self._ctx, capi.COLPIV_HOUSEHOLDER) if rank == root else None
        self._host_comm = dist.get_backend(group) == "gloo"

    def local_layout(self):
        return self.block_rows[self.start:self.end], self.block_cols[self.start:self.end]

    # -- exchange helpers (equal-sized pieces: one n x n triangle or one n-vector per rank)
    def _gather(self, t: torch.Tensor):
        dev = t.device
        assert self._host_comm or dev == self._ctx.device, "RCCL collectives need the tensor on the rank's own GPU"
        t = t.contiguous().cpu() if self._host_comm else t.contiguous()
        lst = [torch.empty_like(t) for _ in range(self.world)] if self.rank_id == self.root else None
        dist.gather(t, lst, dst=self.root, group=self.group)
        return None if lst is None else [p.to(dev) for p in lst]

    def _bcast(self, t: torch.Tensor):
        dev = t.device
        c = t.contiguous().cpu() if self._host_comm else t.contiguous()
        dist.broadcast(c, src=self.root, group=self.group)
        return c.to(dev)

    @staticmethod
    def _colmajor(t: torch.Tensor) -> torch.Tensor:
        return t.t().contiguous().t()

    def compute(self, local_left, local_J2: torch.Tensor):
        dev = self._ctx.device
        m2 = self.m2
        self.m_leftSolver.compute(local_left)
        n1, m1 = local_left.rows(), local_left.cols()
        J2 = local_J2.to(dev, torch.float64)
        assert J2.shape[0] >= n1 and J2.shape[1] == m2
        T = self.m_leftSolver.applyQt(J2[:n1, :].contiguous())
        self._S = T[:m1, :].clone()
        bottom = self._colmajor(torch.cat([T[m1:, :], J2[n1:, :]], dim=0))
        self._n1, self._m1, self._nb = n1, m1, bottom.shape[0]
        if bottom.shape[0] >= m2:
            self._tsqr.compute(bottom)                       # bottom_g = Q0_g R0_g on this GPU
            tri = self._tsqr.matrixR()
            self._reduced = True
        else:                                                # fewer rows than columns: the rows themselves, padded
            tri = torch.zeros((m2, m2), dtype=torch.float64, device=dev)
            tri[:bottom.shape[0], :] = bottom
            self._reduced = False
        parts = self._gather(tri)
        perm = torch.empty(m2, dtype=torch.int32, device=dev)
        if self.rank_id == self.root:
            stack = self._colmajor(torch.cat(parts, dim=0))  # (world n) x n
            self.m_rightSolver.compute(stack)
            perm = self.m_rightSolver.colsPermutation().clone()
        self._P2 = self._bcast(perm).long()
        return self

    def colsPermutationRight(self) -> np.ndarray:
        return self._P2.cpu().numpy()

    def solve(self, b_local: torch.Tensor):
        """Least-squares solution: returns (x1_local, x2): the entries of x that belong to this rank's tiles, and the m2 entries of
        the right block (on every rank).  b_local: this rank's rows of the right-hand side."""
        dev = self._ctx.device
        m2, n1, m1 = self.m2, self._n1, self._m1
        b = b_local.to(dev, torch.float64).reshape(-1, 1)
        y = self.m_leftSolver.applyQt(b[:n1, :].contiguous())
        y1 = y[:m1, :]
        yb = self._colmajor(torch.cat([y[m1:, :], b[n1:, :]], dim=0))
        if self._reduced:
            self._tsqr.applyQ(yb, transpose=True)
            t = yb[:m2, :].clone()
        else:
            t = torch.zeros((m2, 1), dtype=torch.float64, device=dev)
            t[:yb.shape[0], :] = yb
        parts = self._gather(t)
        z2 = torch.empty((m2, 1), dtype=torch.float64, device=dev)
        if self.rank_id == self.root:
            ys = self._colmajor(torch.cat(parts, dim=0))
            self.m_rightSolver.applyQ(ys, transpose=True)
            z2 = self.m_rightSolver.solveR(self._colmajor(ys[:m2, :].clone()))
        z2 = self._bcast(z2)
        # y1 -= S(:, P2) z2 on the device with the strip as it lies there (qrk_dense_gemv_sub), not a library GEMM on a permuted copy
        from . import _capi as capi
        S = self._S if self._S.t().is_contiguous() else self._colmajor(self._S)
        rhs1 = self._colmajor(y1.clone())
        p2 = self._P2.to(torch.int32).contiguous()
        zc = z2[:, 0].contiguous()
        self._ctx.use_current_stream()
        capi.check(capi.lib().qrk_dense_gemv_sub(self._ctx.handle, S.data_ptr(), S.stride(1), m1, m2, p2.data_ptr(), zc.data_ptr(),
                                                 rhs1[:, 0].data_ptr()), self._ctx.handle)
        z1 = self.m_leftSolver.solveR(rhs1)
        p1 = torch.as_tensor(self.m_leftSolver.colsPermutation(), device=dev).long()
        x1 = torch.empty_like(z1)
        x1[p1, :] = z1
        x2 = torch.empty_like(z2)
        x2[self._P2, :] = z2
        return x1[:, 0], x2[:, 0]
