"""Block sharding across the GPUs of one node (one process per GPU, torch.distributed / RCCL).

The diagonal blocks of a SparseBlockDiagonal are independent (the hot loop of
BlockDiagonalSparseQR::factorize, BlockDiagonalSparseQR.h:432, carries only the running offsets
base_row/base_col), so they shard as contiguous ranges with NO data-path collective: every rank
factorises its range and its Q / R / perm shards are already in final global order.  The only
exchange is the optional gather of the composed R (and perm) shards for a caller that needs the
whole factor on every rank (all_gather over RCCL/xGMI; `gloo` in the CPU tests).
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


def all_gather_ragged(local: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenate 1-D shards of different lengths from all ranks, in rank order, on every rank."""
    world = dist.get_world_size(group)
    n = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    padded = torch.zeros(mx, dtype=local.dtype, device=local.device)
    padded[:local.numel()] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)])


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

    def gatherR(self) -> torch.Tensor:
        return all_gather_ragged(self.solver.rValues().contiguous(), self.group)

    def gatherPerm(self) -> torch.Tensor:
        """Global m_outputPerm_c indices: shard-local indices shifted by the shard's base_col."""
        p = torch.as_tensor(self.solver.colsPermutation()).to(self.solver.rValues().device).to(torch.int32)
        return all_gather_ragged((p + self.base_col).contiguous(), self.group)
