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
