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
