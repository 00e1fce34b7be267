"""Shared helpers for the parity tests (oracle side + comparison rules)."""
import numpy as np

from oracle import oracle as orc

# Stated parity rule: permutations / indices bit-exact; Q, R, tau, x within
# 1e-12 relative Frobenius norm for the block-diagonal path (the reference's own bar is 1e-6,
# test/test.h:31).
RTOL = 1e-12


def rel_fro(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    den = min(np.linalg.norm(a), np.linalg.norm(b))
    num = np.linalg.norm(a - b)
    return num / den if den > 0 else num


def seeded_tiles(seed, lo, hi, n):
    """Deterministic tile values: libstdc++ default_random_engine(seed) + uniform_real_distribution(lo,hi),
    the generator of test/test-qrkit.cpp:64-65 (restated in oracle/qrk_oracle.c)."""
    return orc.gen_uniform(seed, lo, hi, n)


def oracle_factorize(rows, cols, tiles, mat_rows=None, q_format=0, block_solver=0):
    prob = orc.BDProblem(rows, cols, tiles, matRows=mat_rows, q_format=q_format, block_solver=block_solver)
    return prob, prob.factorize()


def per_tile_rel(got, want, sizes):
    """Largest per-tile relative Frobenius error of a packed per-tile array (`sizes[i]` values for tile i):
    one tile off by 1e-11 among 1000 good ones fails, which a single batch-wide ratio would hide."""
    got = np.asarray(got, dtype=np.float64).ravel()
    want = np.asarray(want, dtype=np.float64).ravel()
    sizes = np.asarray(sizes, dtype=np.int64)
    assert got.size == want.size == int(sizes.sum()), (got.size, want.size, int(sizes.sum()))
    ends = np.cumsum(sizes)
    starts = ends - sizes
    d2 = np.add.reduceat((got - want) ** 2, starts)[sizes > 0] if got.size else np.zeros(0)
    w2 = np.add.reduceat(want ** 2, starts)[sizes > 0] if got.size else np.zeros(0)
    rel = np.sqrt(d2) / np.where(w2 > 0, np.sqrt(w2), 1.0)
    return float(rel.max()) if rel.size else 0.0


def tile_sizes(rows, cols):
    """(values per tile of Q, of packed R, of tau) for per_tile_rel."""
    r = np.asarray(rows, dtype=np.int64)
    c = np.asarray(cols, dtype=np.int64)
    return r * r, c * (c + 1) // 2, c
