#!/usr/bin/env python3
"""bench.py -- block-diagonal QR factorizations/s on MI355X (BASELINE.json metric).

A "step" is one numeric factorisation (BlockDiagonalSparseQR::factorize,
src/QRKit/BlockDiagonalSparseQR.h:415-547) of one synthetic block-diagonal matrix of
BASELINE configs[1]: 10000 diagonal blocks of 32x32, double, column-pivoted Householder
per block, explicit Q, packed R and the column permutation written to HBM.  Inputs are
resident in HBM before the timed region; steps rotate over several distinct matrices so
that the working set (207 MB per matrix) exceeds the 256 MiB Infinity Cache.

  python bench.py --gpus N --steps K --warmup W

N > 1 works as typed: this process starts N ranks (torch.distributed.run, one per GPU, RCCL) BEFORE it makes any GPU call
and exits with their return code; launched by torch.distributed.run itself (WORLD_SIZE set) it is one of the ranks.

The headline `value` is weak scaling, as the path partitions: every rank factorises its own stream of 10000-block matrices
with no data-path collective; value = all matrices of all ranks / max time.  For N > 1 the same run also measures STRONG
scaling with the only exchange the path has, reported under "strong_scaling": ONE B-block matrix cut into contiguous block
ranges (qrkit_amd/sharding.py), every rank factorises its range, R and the permutation are gathered on rank 0 with their true
byte counts (grouped send/recv over RCCL) INSIDE the timed region; B = 10000 (the BASELINE shape) and B = 1000000.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BLOCKS, BR, BC = 10000, 32, 32
# algorithmic bytes per 32x32 tile = 8192 (read A) + 8192 (write Q) + 4224 (write R) + 128 (perm)
BYTES_PER_TILE = 8 * BR * BC + 8 * BR * BR + 8 * (BC * (BC + 1) // 2) + 4 * BC
FLOPS_PER_TILE = (2 * BR * BC * BC - 2 * BC ** 3 / 3) + 4 * (BR * BR * BC - BR * BC * BC + BC ** 3 / 3)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (CPU restatement of the reference path, kind = "port"), in child processes that never touch a GPU
# ---------------------------------------------------------------------------------------------------------------------
def _cpu_worker(seconds: float):
    """Child process: times orc_bd_factorize on 1000 blocks of 32x32 (BASELINE configs[0]) for `seconds`."""
    from oracle import oracle as orc           # (QRK_ORACLE_LIB, set by cpu_baseline, selects the -O3 -march=native timing copy)
    nb = 1000
    tiles = orc.gen_uniform(1, 0.5, 5.0, nb * BR * BC)
    prob = orc.BDProblem.uniform(nb, BR, BC, tiles)
    prob.factorize()
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < seconds:
        prob.factorize()
        reps += 1
    dt = time.perf_counter() - t0
    print(json.dumps({"reps": reps, "seconds": dt, "tiles_per_s": reps * nb / dt}), flush=True)


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seconds: float):
    """Runs BEFORE this process initialises HIP/RCCL (children are plain subprocesses; nothing is forked from a GPU process)."""
    import tempfile
    fast = os.path.join(tempfile.gettempdir(), f"libqrk_oracle_fast_{os.getpid()}.so")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "fast", f"FAST={fast}"])   # built on THIS host
    env = dict(os.environ, QRK_ORACLE_LIB=fast)

    def spawn(sec):
        return subprocess.Popen([sys.executable, os.path.abspath(__file__), "--_cpu-worker", str(sec)],
                                stdout=subprocess.PIPE, text=True, cwd=ROOT, env=env)

    def result(p):
        out, _ = p.communicate()
        return json.loads(out.strip().splitlines()[-1])

    one = result(spawn(seconds))
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = max(1, min(ncores, 16))          # the GPU box's CPU share for one GPU
    allc = [result(p) for p in [spawn(max(seconds / 2, 3.0)) for _ in range(ncores)]]
    rate_all = sum(r["tiles_per_s"] for r in allc)
    try:
        os.remove(fast)
    except OSError:
        pass
    return {"value": one["tiles_per_s"] / BLOCKS, "unit": "factorizations/s", "cores": 1, "kind": "port",
            "sample": f"{one['reps']} x (1000 blocks of 32x32) in {one['seconds']:.1f} s, oracle/qrk_oracle.c built "
                      "gcc -O3 -march=native (timing copy; the checker copy stays -O2 -ffp-contract=off), single thread "
                      "(the reference's hot loop is single-threaded, BlockDiagonalSparseQR.h:432)",
            "block_factorizations_per_s": one["tiles_per_s"],
            "cpu_model": _cpu_model(),
            "allcores": {"cores": ncores, "value": rate_all / BLOCKS, "block_factorizations_per_s": rate_all,
                         "sample": f"{ncores} independent processes, {allc[0]['seconds']:.1f} s each, same workload"}}


# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher: start the ranks as children (no GPU call has happened here)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def parse(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--sets", type=int, default=8, help="distinct matrices rotated over (working set > L3)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the comparison of 500 tiles with the oracle (outside the timed region)")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the strong-scaling legs")
    ap.add_argument("--no-steady", action="store_true", help="N = 1: skip the 160000-tile steady-state leg (profiling runs)")
    ap.add_argument("--strong-blocks", type=str, default="10000,1000000")
    ap.add_argument("--_cpu-worker", type=float, default=None, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if args._cpu_worker is not None:
        return _cpu_worker(args._cpu_worker)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, argv))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # the CPU leg first: this process has not touched HIP or RCCL yet (rank 0, N = 1 only)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_seconds)

    import numpy as np
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the product has no CPU path")
    backend = os.environ.get("QRK_BENCH_BACKEND", "nccl")     # "gloo": rehearsal of the N > 1 path on a one-GPU box
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or os.environ.get("QRK_BENCH_FORCE_DIST"):   # (the env switch exercises the collective path on one GPU)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import qrkit_amd
    from qrkit_amd import _capi as capi
    ctx = qrkit_amd.Context(dev_index)

    def make_plan(nblocks):
        lay = capi.BDLayout()
        lay.num_blocks, lay.block_rows, lay.block_cols = nblocks, BR, BC
        lay.rows = lay.cols = None
        lay.mat_rows, lay.mat_cols = nblocks * BR, nblocks * BC
        plan = C.c_void_p()
        capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), capi.FULL_Q, capi.COLPIV_HOUSEHOLDER,
                                                 C.byref(plan)), ctx.handle)
        return plan

    plan = make_plan(BLOCKS)
    tl, nq, nr, ncol = BLOCKS * BR * BC, BLOCKS * BR * BR, BLOCKS * (BC * (BC + 1) // 2), BLOCKS * BC
    S = args.sets

    # synthetic data, U(0.5, 5) like the reference tests (test/test-qrkit.cpp:64-65), distinct per set and rank
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    tiles = torch.rand(S * tl, generator=g, device=dev, dtype=torch.float64) * 4.5 + 0.5
    qv = torch.empty(S * nq, device=dev, dtype=torch.float64)
    rv = torch.empty(S * nr, device=dev, dtype=torch.float64)
    pm = torch.empty(S * ncol, device=dev, dtype=torch.int32)

    def run(p, iters, t=tiles, q=qv, r=rv, pp=pm, sets=S):
        ms = C.c_float()
        capi.check(capi.lib().qrk_bd_time_factorize(p, t.data_ptr(), q.data_ptr(), r.data_ptr(), pp.data_ptr(),
                                                    sets, iters, C.byref(ms)), ctx.handle)
        return ms.value

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        run(plan, args.warmup)
    barrier()
    t0 = time.perf_counter()
    kernel_ms = run(plan, args.steps)   # K launches, HIP events on the launch stream around them
    barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([wall, kernel_ms], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, kernel_ms = t[0].item(), t[1].item()

    # ---- outside the timed region: the kernel instantiation that was just timed (tau not stored) against the oracle
    checked = None
    if not args.no_check and rank == 0:
        from oracle import oracle as orc
        nchk = 500
        host = tiles[:nchk * BR * BC].cpu().numpy()
        ref = orc.BDProblem.uniform(nchk, BR, BC, host).factorize()
        capi.check(capi.lib().qrk_bd_factorize(plan, tiles.data_ptr(), qv.data_ptr(), rv.data_ptr(), pm.data_ptr(),
                                               None, capi.MEM_DEVICE), ctx.handle)
        torch.cuda.synchronize()
        if not np.array_equal(pm[:nchk * BC].cpu().numpy(), ref.perm):
            raise SystemExit("bench.py: column permutation differs from the oracle")
        Q = qv[:nchk * BR * BR].cpu().numpy().reshape(nchk, -1)
        R = rv[:nchk * 528].cpu().numpy().reshape(nchk, -1)
        eq = (np.linalg.norm(Q - ref.Q_vals.reshape(nchk, -1), axis=1) / np.linalg.norm(ref.Q_vals.reshape(nchk, -1), axis=1)).max()
        er = (np.linalg.norm(R - ref.R_vals.reshape(nchk, -1), axis=1) / np.linalg.norm(ref.R_vals.reshape(nchk, -1), axis=1)).max()
        if not (eq <= 1e-12 and er <= 1e-12):
            raise SystemExit(f"bench.py: Q/R differ from the oracle: {eq} {er}")
        checked = {"tiles": nchk, "perm": "bit-exact", "max_tile_rel_err_Q": float(eq), "max_tile_rel_err_R": float(er)}

    # ---- steady state (N = 1): the same kernel on 160000 tiles per launch, where the three-round tail of 10000 tiles vanishes
    steady = None
    if world == 1 and not args.no_steady:
        nb_big = 160000
        pbig = make_plan(nb_big)
        tb = torch.rand(nb_big * BR * BC, generator=g, device=dev, dtype=torch.float64) * 4.5 + 0.5
        qb = torch.empty(nb_big * BR * BR, device=dev, dtype=torch.float64)
        rb = torch.empty(nb_big * 528, device=dev, dtype=torch.float64)
        pb = torch.empty(nb_big * BC, device=dev, dtype=torch.int32)
        run(pbig, 5, tb, qb, rb, pb, 1)
        ms_big = run(pbig, 30, tb, qb, rb, pb, 1)
        gbs = BYTES_PER_TILE * nb_big / (ms_big * 1e-3) / 1e9
        steady = {"blocks_per_launch": nb_big, "avg_launch_ms": ms_big, "achieved_GBs": gbs, "frac": gbs / HBM_PEAK_GBS,
                  "block_factorizations_per_s": nb_big / (ms_big * 1e-3)}
        capi.lib().qrk_bd_plan_destroy(pbig)
        del tb, qb, rb, pb

    # ---- strong scaling (N > 1): one B-block matrix over the ranks, R and perm gathered on rank 0 inside the timed region
    strong = None
    if dist is not None and world > 1 and not args.no_strong:
        strong = strong_scaling(args, ctx, dev, dist, backend, rank, world, torch, np)

    if rank == 0:
        fact_per_s = world * args.steps / wall
        bytes_per_launch = BYTES_PER_TILE * BLOCKS
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        capi.lib().qrk_bd_kernel_name.restype = C.c_char_p
        capi.lib().qrk_bd_kernel_name.argtypes = [C.c_void_p, C.c_int]
        kname = capi.lib().qrk_bd_kernel_name(plan, 0).decode()
        tr = _traffic()
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
                       "parallelism": f"{world} rank(s), one per GPU; ranks seen by the process group: "
                                      f"{dist.get_world_size() if dist is not None else 1} ({backend if dist is not None else 'no collective'}); "
                                      "each rank factorises its own stream of matrices, no data-path collective"},
            "block_factorizations_per_s": fact_per_s * BLOCKS,
            "gflops": fact_per_s * BLOCKS * FLOPS_PER_TILE / 1e9,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": tr[0], "traffic_source": tr[1],
                         "kernel": kname, "avg_launch_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": bytes_per_launch},
            "checked": checked if checked is not None else False,
        }
        if steady is not None:
            out["steady_state"] = steady
        if strong is not None:
            out["strong_scaling"] = strong
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)

    capi.lib().qrk_bd_plan_destroy(plan)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def strong_scaling(args, ctx, dev, dist, backend, rank, world, torch, np):
    """One B-block matrix, contiguous block ranges per rank (sharding.shard_ranges), factorise, gather R + perm on rank 0."""
    from qrkit_amd import _capi as capi
    from qrkit_amd.sharding import gather_ragged_to_root, shard_ranges
    res = []
    for B in [int(x) for x in args.strong_blocks.split(",") if x]:
        rows = np.full(B, BR, np.int32)
        ranges = shard_ranges(rows, rows, world)
        lo, hi = ranges[rank]
        nb = hi - lo
        lay = capi.BDLayout()
        lay.num_blocks, lay.block_rows, lay.block_cols = nb, BR, BC
        lay.rows = lay.cols = None
        lay.mat_rows, lay.mat_cols = nb * BR, nb * BC
        plan = C.c_void_p()
        capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), capi.FULL_Q, capi.COLPIV_HOUSEHOLDER, C.byref(plan)),
                   ctx.handle)
        g = torch.Generator(device=dev)
        g.manual_seed(99 + rank)
        t = torch.rand(nb * BR * BC, generator=g, device=dev, dtype=torch.float64) * 4.5 + 0.5
        q = torch.empty(nb * BR * BR, device=dev, dtype=torch.float64)
        r = torch.empty(nb * 528, device=dev, dtype=torch.float64)
        p = torch.empty(nb * BC, device=dev, dtype=torch.int32)
        r_sizes = [(b - a) * 528 for a, b in ranges]
        p_sizes = [(b - a) * BC for a, b in ranges]
        xdev = dev if backend == "nccl" else "cpu"      # (gloo rehearsal: the shards are staged through the host)
        R_all = torch.empty(B * 528, device=xdev, dtype=torch.float64) if rank == 0 else None
        P_all = torch.empty(B * BC, device=xdev, dtype=torch.int32) if rank == 0 else None
        base_col = lo * BC

        def step(gather):
            capi.check(capi.lib().qrk_bd_factorize(plan, t.data_ptr(), q.data_ptr(), r.data_ptr(), p.data_ptr(), None,
                                                   capi.MEM_DEVICE), ctx.handle)
            if gather:
                gather_ragged_to_root(r.to(xdev), r_sizes, R_all, 0, rank, world)
                gather_ragged_to_root((p + base_col).to(xdev), p_sizes, P_all, 0, rank, world)   # global m_outputPerm_c indices

        def timed(iters, gather):
            dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                step(gather)
            torch.cuda.synchronize(); dist.barrier()
            dt = time.perf_counter() - t0
            tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return tt.item() / iters

        iters = 50 if B <= 100000 else 10
        timed(3, True)
        with_g = timed(iters, True)
        without = timed(iters, False)
        res.append({"blocks": B, "blocks_per_rank": [b - a for a, b in ranges], "ms_per_factorization_with_gather": with_g * 1e3,
                    "ms_factorize_only": without * 1e3, "gather_ms": (with_g - without) * 1e3,
                    "factorizations_per_s": 1.0 / with_g, "block_factorizations_per_s": B / with_g,
                    "gathered_bytes_on_root": B * (528 * 8 + BC * 4)})
        capi.lib().qrk_bd_plan_destroy(plan)
        del t, q, r, p, R_all, P_all
    return {"scaling": "strong", "collective": "grouped send/recv of R (f64) and perm (i32) shards to rank 0, true byte counts, "
                                               "inside the timed region", "backend": backend, "runs": res}


def _traffic():
    """HBM bytes per launch from the committed PMC pass (profiles/), or null -- NOT measured in this run."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(p):
        try:
            with open(p) as f:
                d = json.load(f)
            return d.get("hbm_bytes_per_launch"), f"profiles/pmc_traffic.json: committed rocprofv3 --pmc pass ({d.get('round', 'r01')}), not measured in this run"
        except Exception:
            return None, None
    return None, None


if __name__ == "__main__":
    main()
