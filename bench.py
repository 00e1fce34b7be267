#!/usr/bin/env python3
"""bench.py -- block-diagonal QR factorizations/s on MI355X (BASELINE.json metric).

A "step" is one numeric factorisation (BlockDiagonalSparseQR::factorize,
src/QRKit/BlockDiagonalSparseQR.h:415-547) of one synthetic block-diagonal matrix of
BASELINE configs[1]: 10000 diagonal blocks of 32x32, double, column-pivoted Householder
per block, explicit Q, packed R and the column permutation written to HBM.  Inputs are
resident in HBM before the timed region; steps rotate over several distinct matrices so
that the working set (207 MB per matrix) exceeds the 256 MiB Infinity Cache.

  python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run)

N > 1 is weak scaling: every rank factorises its own stream of matrices (blocks shard
across GPUs with no data-path collective); value = all matrices of all ranks / max time.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BLOCKS, BR, BC = 10000, 32, 32
# algorithmic bytes per 32x32 tile = 8192 (read A) + 8192 (write Q) + 4224 (write R) + 128 (perm)
BYTES_PER_TILE = 8 * BR * BC + 8 * BR * BR + 8 * (BC * (BC + 1) // 2) + 4 * BC
FLOPS_PER_TILE = (2 * BR * BC * BC - 2 * BC ** 3 / 3) + 4 * (BR * BR * BC - BR * BC * BC + BC ** 3 / 3)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(seconds: float, allcores: bool):
    """The oracle (CPU restatement of the reference path, kind = "port") timed on the host."""
    from oracle import oracle as orc
    nb = 1000   # BASELINE configs[0]: 1000 blocks of 32x32
    tiles = orc.gen_uniform(1, 0.5, 5.0, nb * BR * BC)
    prob = orc.BDProblem.uniform(nb, BR, BC, tiles)
    prob.factorize()   # warm
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < seconds:
        prob.factorize()
        reps += 1
    dt = time.perf_counter() - t0
    tiles_per_s = reps * nb / dt
    out = {"value": tiles_per_s / BLOCKS, "unit": "factorizations/s", "cores": 1, "kind": "port",
           "sample": f"{reps} x (1000 blocks of 32x32) in {dt:.1f} s, oracle/qrk_oracle.c, gcc -O2, single thread "
                     "(the reference's hot loop is single-threaded, BlockDiagonalSparseQR.h:432)",
           "block_factorizations_per_s": tiles_per_s}
    if allcores:
        import multiprocessing as mp
        n = os.cpu_count() or 1
        with mp.get_context("fork").Pool(n) as pool:
            res = pool.map(_cpu_worker, [max(seconds / 2, 2.0)] * n)
        out["allcores"] = {"cores": n, "value": sum(res) / BLOCKS, "block_factorizations_per_s": sum(res)}
    return out


def _cpu_worker(seconds):
    from oracle import oracle as orc
    nb = 1000
    tiles = orc.gen_uniform(1, 0.5, 5.0, nb * BR * BC)
    prob = orc.BDProblem.uniform(nb, BR, BC, tiles)
    prob.factorize()
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < seconds:
        prob.factorize()
        reps += 1
    return reps * nb / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--sets", type=int, default=8, help="distinct matrices rotated over (working set > L3)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-allcores", action="store_true")
    ap.add_argument("--check", action="store_true", help="verify one matrix against the oracle before timing")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("QRK_BENCH_FORCE_DIST"):   # (the env switch exercises the RCCL path on one GPU)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import qrkit_amd
    from qrkit_amd import _capi as capi
    ctx = qrkit_amd.Context(local_rank)
    lay = capi.BDLayout()
    lay.num_blocks, lay.block_rows, lay.block_cols = BLOCKS, BR, BC
    lay.rows = lay.cols = None
    lay.mat_rows, lay.mat_cols = BLOCKS * BR, BLOCKS * BC
    plan = C.c_void_p()
    capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), capi.FULL_Q, capi.COLPIV_HOUSEHOLDER,
                                             C.byref(plan)), ctx.handle)
    tl, nq, nr = C.c_int64(), C.c_int64(), C.c_int64()
    capi.check(capi.lib().qrk_bd_plan_sizes(plan, C.byref(tl), C.byref(nq), C.byref(nr)), ctx.handle)
    tl, nq, nr = tl.value, nq.value, nr.value
    S = args.sets

    # synthetic data, U(0.5, 5) like the reference tests (test/test-qrkit.cpp:64-65), distinct per set and rank
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    tiles = torch.rand(S * tl, generator=g, device=dev, dtype=torch.float64) * 4.5 + 0.5
    qv = torch.empty(S * nq, device=dev, dtype=torch.float64)
    rv = torch.empty(S * nr, device=dev, dtype=torch.float64)
    pm = torch.empty(S * lay.mat_cols, device=dev, dtype=torch.int32)

    def run(iters):
        ms = C.c_float()
        capi.check(capi.lib().qrk_bd_time_factorize(plan, tiles.data_ptr(), qv.data_ptr(), rv.data_ptr(), pm.data_ptr(),
                                                    S, iters, C.byref(ms)), ctx.handle)
        return ms.value

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    kernel_ms = run(args.steps)   # K launches, HIP events on the launch stream around them
    barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([wall, kernel_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, kernel_ms = t[0].item(), t[1].item()

    if args.check and rank == 0:
        from oracle import oracle as orc
        nchk = 500
        host = tiles[:nchk * BR * BC].cpu().numpy()
        ref = orc.BDProblem.uniform(nchk, BR, BC, host).factorize()
        capi.check(capi.lib().qrk_bd_factorize(plan, tiles.data_ptr(), qv.data_ptr(), rv.data_ptr(), pm.data_ptr(),
                                               None, capi.MEM_DEVICE), ctx.handle)
        torch.cuda.synchronize()
        assert np.array_equal(pm[:nchk * BC].cpu().numpy(), ref.perm)
        eq = np.linalg.norm(qv[:nchk * BR * BR].cpu().numpy() - ref.Q_vals) / np.linalg.norm(ref.Q_vals)
        er = np.linalg.norm(rv[:nchk * 528].cpu().numpy() - ref.R_vals) / np.linalg.norm(ref.R_vals)
        assert eq < 1e-12 and er < 1e-12, (eq, er)

    if rank == 0:
        fact_per_s = world * args.steps / wall
        bytes_per_launch = BYTES_PER_TILE * BLOCKS
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "block-diagonal QR factorizations/sec",
            "value": fact_per_s,
            "unit": "factorizations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "10000 blocks of 32x32 block-diagonal, double (BASELINE configs[1]); one step = "
                                   "factorize() of one such matrix: per-block ColPivHouseholderQR, explicit Q, packed R, "
                                   "column permutation; FullQ format",
                       "blocks": BLOCKS, "block_rows": BR, "block_cols": BC, "matrices_rotated": S,
                       "parallelism": f"{world} independent block shards (one matrix stream per GPU), no collective"},
            "block_factorizations_per_s": fact_per_s * BLOCKS,
            "gflops": fact_per_s * BLOCKS * FLOPS_PER_TILE / 1e9,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": _traffic(),
                         "kernel": "qrk::bdqr_pair32_kernel<true, false>", "avg_launch_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": bytes_per_launch},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, args.cpu_allcores)
        print(json.dumps(out), flush=True)

    capi.lib().qrk_bd_plan_destroy(plan)
    if dist is not None:
        dist.destroy_process_group()


def _traffic():
    """HBM bytes per launch from the committed PMC pass (profiles/), or null."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(p):
        try:
            with open(p) as f:
                return json.load(f).get("hbm_bytes_per_launch")
        except Exception:
            return None
    return None


if __name__ == "__main__":
    main()
