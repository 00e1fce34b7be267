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
    """Each rank factorises its contiguous block range; gatherR()/gatherPerm() compose the global factor, solve() answers the least-squares
    problem with x as the only thing that travels, computeGatherR() hides the gather of R behind the factorisation of the next piece.

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
            # one process per GPU: the rank's own device (LOCAL_RANK under torch.distributed.run), never "device 0 on every rank"; one C-ABI
            # handle for the solver and for the pieces of computeGatherR
            import os
            from .solvers import BlockDiagonalSparseQR, Context
            ndev = max(torch.cuda.device_count(), 1)
            ctx = Context(int(os.environ.get("LOCAL_RANK", rank)) % ndev)
            solver_factory = lambda: BlockDiagonalSparseQR(context=ctx)
        self._solver_factory = solver_factory
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


    # -- consumers that scale: the gather of R is the bound of the strong-scaling legs (4 352 B per 32 x 32 tile over one xGMI link per
    #    sender, bench.py strong_shard_emulation), so (1) a least-squares consumer gathers x only and (2) a consumer that does need R
    #    sends a chunk while the next one is being factorised
    def solve(self, b_local, root=None) -> torch.Tensor:
        """x = solve(J, b) without the composed R.  _solve_impl (BlockDiagonalSparseQR.h:257-280) is block-local -- Q, R and the column
        permutation are block diagonal -- so the rank solves its own range on its own factors (b_local: the rows of b that belong
        to its blocks; [rows_g] or [rows_g, nrhs]) and ONLY x travels: 8 bytes per column and right-hand side (256 B per 32 x 32 tile
        against 4 352 B of R and permutation).  Returns x of the whole matrix ([cols] or [cols, nrhs]) on every rank (root=None) or on
        `root` (None elsewhere)."""
        x = self.solver.solve(b_local)
        x = torch.as_tensor(x)
        vec = x.dim() == 1
        x2 = x.reshape(x.shape[0], -1).contiguous()                  # [cols_g, nrhs] row-major: shards concatenate along dim 0
        nrhs = x2.shape[1]
        sizes = [n * nrhs for n in self._c_sizes()]
        flat = x2.reshape(-1)
        if root is None:
            out = all_gather_ragged(flat, self.group, sizes)
        else:
            out = torch.empty(sum(sizes), dtype=flat.dtype, device=flat.device) if self.rank_id == root else None
            gather_ragged_to_root(flat, sizes, out, root, self.rank_id, self.world, self.group)
            if out is None:
                return None
        out = out.reshape(-1, nrhs)
        return out[:, 0] if vec else out

    def chunk_ranges(self, chunks: int) -> List[Tuple[int, int]]:
        """The rank's block range cut into `chunks` contiguous pieces balanced like the shards themselves (global block indices)."""
        lr, lc = self.local_layout()
        return [(self.start + a, self.start + b) for a, b in shard_ranges(lr, lc, max(1, int(chunks)))]

    def computeGatherR(self, local_mat, chunks: int = 4, root: int = 0, slice_mat=None, solver_factory=None):
        """compute() + gatherR(root) with the exchange hidden behind the arithmetic: the rank's range is cut into `chunks` pieces, each
        with its own plan; piece i is factorised and its R posted (non-blocking grouped send / receive, true byte counts) while piece
        i + 1 is being factorised -- over RCCL the transfers run on the communicator's stream beside the kernels of the next piece.
        Every rank uses the same number of pieces (the root posts one receive per peer and piece).  Returns the composed packed R on
        `root` (None elsewhere); afterwards rValues() / colsPermutation() / solve() of this object answer from the pieces.
        slice_mat(local_mat, a, b): the blocks [a, b) of the rank's matrix (default: SparseBlockDiagonal tiles, host or device)."""
        chunks = max(1, int(chunks))
        if slice_mat is None:
            slice_mat = _slice_block_diagonal
        if solver_factory is None:
            solver_factory = self._solver_factory
        c = self.block_cols.astype(np.int64)
        r_len = lambda a, b: int((c[a:b] * (c[a:b] + 1) // 2).sum())
        pieces = [ShardedBlockDiagonalQR.chunk_ranges(_Range(self.block_rows, self.block_cols, a, b), chunks) for a, b in self.ranges]
        mine = pieces[self.rank_id]
        sizes = self._r_sizes()
        out = None
        solvers, reqs, keep = [], [], []
        for i, (a, b) in enumerate(mine):
            sv = solver_factory()
            if b > a:
                sv.compute(slice_mat(local_mat, a - self.start, b - self.start))
            solvers.append((a, b, sv))
            if self.rank_id == root:
                if out is None:
                    ref = sv.rValues() if b > a else None
                    dev = ref.device if ref is not None else "cpu"
                    out = torch.empty(sum(sizes), dtype=torch.float64, device=dev)
                ops = []
                for peer in range(self.world):
                    pa, pb = pieces[peer][i]
                    off = r_len(0, pa)
                    n = r_len(pa, pb)
                    if n == 0:
                        continue
                    if peer == root:
                        out[off:off + n].copy_(sv.rValues())
                    else:
                        ops.append(dist.P2POp(dist.irecv, out[off:off + n], peer, self.group))
            else:
                ops = []
                if b > a:
                    src = sv.rValues().contiguous()
                    if out is not None and src.device != out.device:
                        src = src.to(out.device)
                    keep.append(src)
                    ops.append(dist.P2POp(dist.isend, src, root, self.group))
            if ops and self.world > 1:
                reqs.extend(dist.batch_isend_irecv(ops))          # (not waited for here: the next piece is factorised meanwhile)
        for q in reqs:
            q.wait()
        self.solver = _ChunkedSolver(solvers, self.block_rows, self.block_cols, self.start)
        return out if self.rank_id == root else None


class _Range:
    """(helper of computeGatherR) the layout fields chunk_ranges reads, for any rank's block range"""

    def __init__(self, block_rows, block_cols, start, end):
        self.block_rows, self.block_cols, self.start, self.end = block_rows, block_cols, start, end

    def local_layout(self):
        return self.block_rows[self.start:self.end], self.block_cols[self.start:self.end]


def _slice_block_diagonal(mat, a: int, b: int):
    """Blocks [a, b) of a SparseBlockDiagonal (host or device tiles) as a SparseBlockDiagonal of its own: a view, no copy."""
    from .solvers import SparseBlockDiagonal
    sz = mat.block_rows.astype(np.int64) * mat.block_cols.astype(np.int64)
    t0, t1 = int(sz[:a].sum()), int(sz[:b].sum())
    tiles = mat.tiles_dev[t0:t1] if mat.tiles_dev is not None else mat.tiles[t0:t1]
    return SparseBlockDiagonal.fromTiles(mat.block_rows[a:b], mat.block_cols[a:b], tiles)


class _ChunkedSolver:
    """The solvers of the pieces of computeGatherR behind the interface ShardedBlockDiagonalQR uses (rValues, qValues,
    colsPermutation, solve): concatenations in block order, local column indices re-based to the rank's range."""

    def __init__(self, pieces, block_rows, block_cols, start):
        self.pieces = [(a, b, sv) for a, b, sv in pieces if b > a]
        self._rows = np.asarray(block_rows, dtype=np.int64)
        self._cols = np.asarray(block_cols, dtype=np.int64)
        self._start = start

    def rValues(self):
        return torch.cat([sv.rValues().reshape(-1) for _, _, sv in self.pieces])

    def qValues(self):
        return torch.cat([sv.qValues().reshape(-1) for _, _, sv in self.pieces])

    def colsPermutation(self):
        out = []
        for a, _, sv in self.pieces:
            out.append(np.asarray(sv.colsPermutation()) + int(self._cols[self._start:a].sum()))
        return np.concatenate(out) if out else np.zeros(0, np.int32)

    def solve(self, b):
        bt = torch.as_tensor(b)
        xs, r0 = [], 0
        for a, e, sv in self.pieces:
            nr = int(self._rows[a:e].sum())
            xs.append(torch.as_tensor(sv.solve(bt[r0:r0 + nr])))
            r0 += nr
        return torch.cat(xs, dim=0)


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
