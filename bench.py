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
    """Child process: times orc_bd_factorize on 1000 blocks of 32x32 (BASELINE configs[0]) for `seconds`.
    QRK_BENCH_FAITHFUL=1: the "faithful assembly" variant, which also reproduces the reference's per-element sparse insertion and
    triplet sort (BlockDiagonalSparseQR.h:424,457-479,536-541)."""
    from oracle import oracle as orc           # (QRK_ORACLE_LIB, set by cpu_baseline, selects the -O3 -march=native timing copy)
    nb = 1000
    faithful = os.environ.get("QRK_BENCH_FAITHFUL") == "1"
    tiles = orc.gen_uniform(1, 0.5, 5.0, nb * BR * BC)
    prob = orc.BDProblem.uniform(nb, BR, BC, tiles)
    prob.factorize(faithful)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < seconds:
        prob.factorize(faithful)
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
    pf = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--_cpu-worker", str(max(seconds / 3, 2.0))],
                          stdout=subprocess.PIPE, text=True, cwd=ROOT, env=dict(env, QRK_BENCH_FAITHFUL="1"))
    faithful = result(pf)
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
            "faithful_assembly": {"cores": 1, "value": faithful["tiles_per_s"] / BLOCKS, "block_factorizations_per_s": faithful["tiles_per_s"],
                                  "sample": f"{faithful['reps']} x (1000 blocks of 32x32) in {faithful['seconds']:.1f} s, with the reference's per-element "
                                            "sparse insertion and triplet sort of Q and R (BlockDiagonalSparseQR.h:424,457-479,536-541)"},
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
    ap.add_argument("--batches", type=int, default=25, help="timed batches of exactly --steps steps each; the line reports the median batch")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the comparison of 500 tiles with the oracle (outside the timed region)")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the strong-scaling legs")
    ap.add_argument("--no-steady", action="store_true", help="N = 1: skip the 160000-tile steady-state leg (profiling runs)")
    ap.add_argument("--strong-blocks", type=str, default="10000,1000000")
    ap.add_argument("--mixed-blocks", type=int, default=100000, help="N > 1: tiles of the configs[4] strong leg (mixed 8..256; 0 = skip)")
    ap.add_argument("--angular-blocks", type=int, default=20000, help="N > 1: tiles of the configs[3] strong leg (8x6 + 2000 dense columns; 0 = skip)")
    ap.add_argument("--no-e2e", action="store_true", help="N = 1: skip the host-buffer (PCIe-inclusive) figure")
    ap.add_argument("--no-other", action="store_true", help="N = 1: skip the timing of BASELINE configs[4]'s share of one GPU (12 500 mixed tiles)")
    ap.add_argument("--_cpu-worker", type=float, default=None, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if args._cpu_worker is not None:
        return _cpu_worker(args._cpu_worker)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the CPU leg in the parent, before any rank (or GPU call) exists; rank 0 picks the result up from a file
        if not args.no_cpu_baseline:
            import tempfile
            fd, path = tempfile.mkstemp(prefix="qrk_bench_cpu_", suffix=".json")
            with os.fdopen(fd, "w") as f:
                json.dump(cpu_baseline(args.cpu_seconds), f)
            os.environ["QRK_BENCH_CPU_JSON"] = path
        rc = spawn_ranks(args, argv)
        if os.environ.get("QRK_BENCH_CPU_JSON"):
            try:
                os.remove(os.environ["QRK_BENCH_CPU_JSON"])
            except OSError:
                pass
        sys.exit(rc)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # the CPU leg first: this process has not touched HIP or RCCL yet (rank 0, N = 1 only)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_seconds)
    elif rank == 0 and os.environ.get("QRK_BENCH_CPU_JSON"):
        try:
            with open(os.environ["QRK_BENCH_CPU_JSON"]) as f:
                cpu = json.load(f)                       # measured by the parent before it started the ranks
        except (OSError, ValueError):
            cpu = None

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
    # A timed batch = EXACTLY --steps steps between barrier + synchronize on both sides, MAX over ranks.  K = 20 steps are 1.7 ms, too
    # short for one sample to be robust (clock ramps, a neighbour on the host), so the batch is repeated --batches times back to
    # back and the line reports the MEDIAN batch (every batch is in "timing").
    walls, kmss = [], []
    for _ in range(max(1, args.batches)):
        barrier()
        t0 = time.perf_counter()
        kms = run(plan, args.steps)     # K launches, HIP events on the launch stream around them
        barrier()
        w = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([w, kms], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            w, kms = t[0].item(), t[1].item()
        walls.append(w); kmss.append(kms)
    order = sorted(range(len(walls)), key=lambda i: walls[i])
    mid = order[len(order) // 2]
    wall, kernel_ms = walls[mid], kmss[mid]
    timing = {"batches": len(walls), "steps_per_batch": args.steps, "statistic": "median batch (wall clock around exactly --steps steps)",
              "batch_ms_per_step_min_median_max": [min(walls) / args.steps * 1e3, wall / args.steps * 1e3, max(walls) / args.steps * 1e3],
              "kernel_us_min_median_max": [min(kmss) * 1e3, sorted(kmss)[len(kmss) // 2] * 1e3, max(kmss) * 1e3],
              "first_batch_ms_per_step": walls[0] / args.steps * 1e3,
              "note": "value / ms_per_step are the MEDIAN batch since round 4; rounds 1-3 reported one batch of --steps steps: "
                      "first_batch_ms_per_step is that statistic for like-for-like comparison across rounds",
              "timed_region_ms_total": sum(walls) * 1e3}

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
        if args.mixed_blocks > 0:
            strong["configs4_mixed"] = strong_mixed(args, ctx, dev, dist, backend, rank, world, torch, np)
        if args.angular_blocks > 0:
            strong["configs3_angular"] = strong_angular(args, ctx, dev, dist, backend, rank, world, torch, np)

    # ---- another BASELINE configuration on this GPU (N = 1; reported next to the headline, never as `value`)
    other = None
    if world == 1 and not args.no_other:
        # (auxiliary legs: a failure in one of them is reported in its place and does not cost the line)
        def leg(fn, *a):
            try:
                return fn(*a)
            except Exception as e:          # noqa: BLE001
                torch.cuda.synchronize()
                return {"error": f"{type(e).__name__}: {e}"}
        other = {"concurrent_streams": leg(concurrent_streams, dev, torch, tiles, qv, rv, pm, S),
                 "strong_shard_emulation": leg(strong_shard_emulation, ctx, dev, torch, np, make_plan, run),
                 "small_tiles": leg(small_tiles, ctx, dev, torch, np),
                 "solve": leg(solve_leg, ctx, dev, torch, np),
                 "configs4_share_of_one_gpu": leg(mixed_share, ctx, dev, torch, np),
                 "configs2_strips": leg(strips_config2, ctx, dev, torch, np),
                 "configs3_angular": leg(angular_config3, ctx, dev, torch, np)}

    # ---- end to end with host buffers (N = 1): tiles over PCIe in, Q / R / perm back (never the headline value)
    e2e = None
    if world == 1 and not args.no_e2e:
        ht = tiles[:tl].cpu().numpy()
        hq = np.empty(nq); hr = np.empty(nr); hp = np.empty(ncol, np.int32)
        lib = capi.lib()
        for _ in range(2):
            capi.check(lib.qrk_bd_factorize(plan, ht.ctypes.data, hq.ctypes.data, hr.ctypes.data, hp.ctypes.data, None, capi.MEM_HOST), ctx.handle)
        t0e = time.perf_counter()
        for _ in range(5):
            capi.check(lib.qrk_bd_factorize(plan, ht.ctypes.data, hq.ctypes.data, hr.ctypes.data, hp.ctypes.data, None, capi.MEM_HOST), ctx.handle)
        dte = (time.perf_counter() - t0e) / 5
        e2e = {"ms_per_factorization": dte * 1e3, "factorizations_per_s": 1.0 / dte,
               "what": "qrk_bd_factorize with QRK_MEM_HOST: pageable host tiles in (82 MB), Q / R / perm out (125 MB) over PCIe + the kernel; "
                       "not the headline value (inputs there are resident in HBM)"}

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
                                      f"GPUs visible to this rank: {ndev}; "
                                      "each rank factorises its own stream of matrices, no data-path collective"},
            "block_factorizations_per_s": fact_per_s * BLOCKS,
            "gflops": fact_per_s * BLOCKS * FLOPS_PER_TILE / 1e9,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": tr[0], "traffic_source": tr[1],
                         "kernel": kname, "avg_launch_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         # what the memory system gives THIS access pattern: a copy-only build of the same kernel (same grid, same loads and
                         # stores, no arithmetic) -- a committed measurement, not part of this run
                         "pattern_floor": {"copy_only_ms_per_launch": [0.0523, 0.0546], "copy_only_GBs": [3800, 3965],
                                           "valu_issue_floor_ms": 0.050,
                                           "source": "profiles/r06_k1_ramp.txt, profiles/r06_k1_pmc_summary.txt (round 6; 10 000 tiles): the "
                                                     "kernel's own loads and stores without its arithmetic, and 4 581 VALU instructions per pair "
                                                     "x 2.24 ns x 4.88 pairs per SIMD; not measured in this run"}},
            "checked": checked if checked is not None else False,
            "timing": timing,
        }
        if steady is not None:
            out["steady_state"] = steady
        if strong is not None:
            out["strong_scaling"] = strong
        if e2e is not None:
            out["end_to_end_host_buffers"] = e2e
        if other is not None:
            out["other_configs"] = other
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

        # the two consumers that scale (qrkit_amd/sharding.py): x only gathered (block-local solve), and R gathered in PIECES pieces,
        # a piece on its way while the next one is factorised
        PIECES = 4
        bvec = torch.rand(nb * BR, generator=g, device=dev, dtype=torch.float64)
        xloc = torch.empty(nb * BC, device=dev, dtype=torch.float64)
        X_all = torch.empty(B * BC, device=xdev, dtype=torch.float64) if rank == 0 else None

        def step_x():
            capi.check(capi.lib().qrk_bd_factorize(plan, t.data_ptr(), q.data_ptr(), r.data_ptr(), p.data_ptr(), None,
                                                   capi.MEM_DEVICE), ctx.handle)
            capi.check(capi.lib().qrk_bd_solve(plan, q.data_ptr(), r.data_ptr(), p.data_ptr(), bvec.data_ptr(), 1, xloc.data_ptr(),
                                               capi.MEM_DEVICE), ctx.handle)
            gather_ragged_to_root(xloc.to(xdev), p_sizes, X_all, 0, rank, world)

        cuts = [[a + (b - a) * k // PIECES for k in range(PIECES + 1)] for a, b in ranges]      # every rank's piece boundaries
        pplans = []
        for k in range(PIECES):
            n = cuts[rank][k + 1] - cuts[rank][k]
            pl = capi.BDLayout()
            pl.num_blocks, pl.block_rows, pl.block_cols = n, BR, BC
            pl.rows = pl.cols = None
            pl.mat_rows, pl.mat_cols = n * BR, n * BC
            pp_ = C.c_void_p()
            if n > 0:
                capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(pl), capi.FULL_Q, capi.COLPIV_HOUSEHOLDER, C.byref(pp_)), ctx.handle)
            pplans.append((pp_, cuts[rank][k] - lo, n))

        def step_ov():
            reqs = []
            for k, (pp_, off, n) in enumerate(pplans):
                if n > 0:
                    capi.check(capi.lib().qrk_bd_factorize(pp_, t.data_ptr() + off * BR * BC * 8, q.data_ptr() + off * BR * BR * 8,
                                                           r.data_ptr() + off * 528 * 8, p.data_ptr() + off * BC * 4, None, capi.MEM_DEVICE),
                               ctx.handle)
                ops = []
                if rank == 0:
                    if n > 0:
                        R_all[(cuts[0][k]) * 528:(cuts[0][k] + n) * 528].copy_(r[off * 528:(off + n) * 528])
                    for peer in range(1, world):
                        a, b = cuts[peer][k], cuts[peer][k + 1]
                        if b > a:
                            ops.append(dist.P2POp(dist.irecv, R_all[a * 528:b * 528], peer))
                elif n > 0:
                    ops.append(dist.P2POp(dist.isend, r[off * 528:(off + n) * 528].to(xdev), 0))
                if ops:
                    reqs.extend(dist.batch_isend_irecv(ops))      # (not waited for: the next piece is factorised meanwhile)
            for w in reqs:
                w.wait()
            gather_ragged_to_root((p + base_col).to(xdev), p_sizes, P_all, 0, rank, world)

        def timed_fn(iters, fn):
            dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                fn()
            torch.cuda.synchronize(); dist.barrier()
            dt = time.perf_counter() - t0
            tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return tt.item() / iters

        iters = 50 if B <= 100000 else 10
        timed(3, True)
        with_g = timed(iters, True)
        without = timed(iters, False)
        timed_fn(3, step_x)
        with_x = timed_fn(iters, step_x)
        timed_fn(3, step_ov)
        with_ov = timed_fn(iters, step_ov)
        res.append({"blocks": B, "blocks_per_rank": [b - a for a, b in ranges], "ms_per_factorization_with_gather": with_g * 1e3,
                    "ms_factorize_only": without * 1e3, "gather_ms": (with_g - without) * 1e3,
                    "factorizations_per_s": 1.0 / with_g, "block_factorizations_per_s": B / with_g,
                    "gathered_bytes_on_root": B * (528 * 8 + BC * 4),
                    "ms_factorize_solve_gather_x_only": with_x * 1e3, "x_bytes_on_root": B * BC * 8,
                    "ms_factorize_with_gather_overlapped": with_ov * 1e3, "pieces": PIECES})
        capi.lib().qrk_bd_plan_destroy(plan)
        for pp_, _, n in pplans:
            if n > 0:
                capi.lib().qrk_bd_plan_destroy(pp_)
        del t, q, r, p, R_all, P_all, X_all
    return {"scaling": "strong", "collective": "grouped send/recv of R (f64) and perm (i32) shards to rank 0, true byte counts, "
                                               "inside the timed region; also timed: the least-squares consumer (block-local solve, x only "
                                               "gathered) and the gather of R in pieces behind the factorisation of the next piece",
            "backend": backend, "runs": res}


def _timed(dist, torch, dev, backend, fn, iters):
    dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize(); dist.barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return tt.item() / iters


def strong_shard_emulation(ctx, dev, torch, np, make_plan, run):
    """What ONE GPU would do in the strong-scaling legs (B tiles of 32 x 32 cut into N contiguous shards), timed on this
    one GPU for N in {1, 2, 4, 8} of B = 10 000 and B = 1 000 000, with the bytes each consumer moves to rank 0 priced at one xGMI
    link per sender (153 GB/s nominal), and the speed-up these imply at best.  Three consumers:
      * R gathered (ShardedBlockDiagonalQR.gatherR): T(B / N) + the R / perm bytes of one shard over its link;
      * R gathered, overlapped (computeGatherR, 4 pieces per rank): the shard factorised as 4 launches of B / (4 N) tiles (timed here),
        a piece's R on its way during the next piece -- max(factorisation, gather) + the smaller of the two / 4;
      * x only (ShardedBlockDiagonalQR.solve): factorisation + block-local solve of the shard (both timed here) + 256 B per tile.
    UNMEASURED on more than one GPU: upper bounds from the kernels' own latency laws, not a scaling result."""
    from qrkit_amd import _capi as capi
    XGMI_LINK_GBS = 153.0
    PIECES = 4
    out = []

    def solve_us(plan, nb, q, r, p, iters):
        b = torch.rand(nb * BR, device=dev, dtype=torch.float64)
        x = torch.empty(nb * BC, device=dev, dtype=torch.float64)
        go = lambda: capi.check(capi.lib().qrk_bd_solve(plan, q.data_ptr(), r.data_ptr(), p.data_ptr(), b.data_ptr(), 1, x.data_ptr(),
                                                         capi.MEM_DEVICE), ctx.handle)
        for _ in range(3):
            go()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            go()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters

    for B, iters in ((10000, 200), (1000000, 6)):
        per, per_solve, per_pieces = {}, {}, {}
        for N in (1, 2, 4, 8):
            nb = B // N
            plan = make_plan(nb)
            S = max(1, min(8, 80000 // nb)) if nb < 80000 else 1
            g = torch.Generator(device=dev); g.manual_seed(4321 + N)
            t = torch.rand(S * nb * BR * BC, generator=g, device=dev, dtype=torch.float64) * 4.5 + 0.5
            q = torch.empty(S * nb * BR * BR, device=dev, dtype=torch.float64)
            r = torch.empty(S * nb * 528, device=dev, dtype=torch.float64)
            p = torch.empty(S * nb * BC, device=dev, dtype=torch.int32)
            run(plan, max(2, iters // 10), t, q, r, p, S)
            per[N] = min(run(plan, iters, t, q, r, p, S) for _ in range(3)) * 1e3      # us per launch
            per_solve[N] = solve_us(plan, nb, q, r, p, max(3, iters // 4))
            capi.lib().qrk_bd_plan_destroy(plan)
            # the shard as PIECES launches of nb / PIECES tiles, one after the other on the stream
            npc = nb // PIECES
            pplan = make_plan(npc)
            run(pplan, PIECES * 2, t, q, r, p, PIECES)
            per_pieces[N] = min(run(pplan, PIECES * max(2, iters // 4), t, q, r, p, PIECES) for _ in range(3)) * 1e3 * PIECES
            capi.lib().qrk_bd_plan_destroy(pplan)
            del t, q, r, p
        legs = []
        for N in (2, 4, 8):
            nb = B // N
            r_bytes = nb * (528 * 8 + BC * 4)                        # one sender's R (f64) + perm (i32)
            x_bytes = nb * BC * 8                                    # one sender's x, one right-hand side
            gather_us = r_bytes / (XGMI_LINK_GBS * 1e3)             # every sender on its own link to rank 0, concurrently
            x_us = x_bytes / (XGMI_LINK_GBS * 1e3)
            t_r = per[N] + gather_us
            t_ov = max(per_pieces[N], gather_us) + min(per_pieces[N], gather_us) / PIECES
            t_x = per[N] + per_solve[N] + x_us
            legs.append({"gpus": N, "tiles_per_gpu": nb, "shard_us": per[N], "shard_in_%d_pieces_us" % PIECES: per_pieces[N],
                         "shard_solve_us": per_solve[N],
                         "gather_bytes_to_rank0": (B - nb) * (528 * 8 + BC * 4), "gather_us_at_one_link_per_sender": gather_us,
                         "x_bytes_to_rank0": (B - nb) * BC * 8, "x_us_at_one_link_per_sender": x_us,
                         "speedup_upper_bound_no_gather": per[1] / per[N],
                         "speedup_upper_bound_with_gather": per[1] / t_r,
                         "speedup_upper_bound_with_gather_overlapped": per[1] / t_ov,
                         "speedup_upper_bound_x_only": (per[1] + per_solve[1]) / t_x})
        out.append({"blocks": B, "one_gpu_us": per[1], "one_gpu_solve_us": per_solve[1], "legs": legs})
    return {"what": "one-GPU emulation of the per-GPU shard of the strong-scaling legs (unmeasured on 8 GPUs): the shard's launch and solve "
                    "times on this GPU, the bytes of the three consumers (R gathered / R gathered overlapped with the factorisation of the "
                    "next piece / x only) and the speed-ups they bound",
            "xgmi_link_GBs_nominal": XGMI_LINK_GBS, "pieces": PIECES, "runs": out}


def concurrent_streams(dev, torch, tiles, qv, rv, pm, S, steps=300):
    """The headline workload (a stream of independent 10 000-tile matrices, distinct buffers per matrix in flight) through ONE handle --
    what `value` reports: every launch waits for the previous one to drain, so the 20-us chain of its last pairs is paid per launch --
    against TWO and THREE handles on their own HIP streams, alternating: the tail of one launch runs beside the head of the next.  Wall
    clock around `steps` factorisations.  What a caller with independent matrices gets by creating one handle per stream; never `value`
    (there a launch is measured alone, and `roofline` prices that launch)."""
    import qrkit_amd
    from qrkit_amd import _capi as capi
    lib = capi.lib()
    res = {}
    for n in (1, 2, 3):
        lanes = []
        for _ in range(n):
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                c = qrkit_amd.Context(dev.index)                 # (bound to the stream that is current here)
            lay = capi.BDLayout()
            lay.num_blocks, lay.block_rows, lay.block_cols = BLOCKS, BR, BC
            lay.rows = lay.cols = None
            lay.mat_rows, lay.mat_cols = BLOCKS * BR, BLOCKS * BC
            plan = C.c_void_p()
            capi.check(lib.qrk_bd_plan_create(c.handle, C.byref(lay), capi.FULL_Q, capi.COLPIV_HOUSEHOLDER, C.byref(plan)), c.handle)
            lanes.append((st, c, plan))

        def step(it):
            _, c, plan = lanes[it % n]
            k = it % S
            capi.check(lib.qrk_bd_factorize(plan, tiles.data_ptr() + 8 * k * BLOCKS * BR * BC, qv.data_ptr() + 8 * k * BLOCKS * BR * BR,
                                            rv.data_ptr() + 8 * k * BLOCKS * 528, pm.data_ptr() + 4 * k * BLOCKS * BC, None, capi.MEM_DEVICE), c.handle)
        for it in range(20):
            step(it)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for it in range(steps):
                step(it)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / steps)
        us = sorted(ts)[1] * 1e6
        res[str(n)] = {"us_per_factorization": us, "factorizations_per_s": 1e6 / us,
                       "algorithmic_GBs": BYTES_PER_TILE * BLOCKS / us / 1e3, "frac_of_hbm": BYTES_PER_TILE * BLOCKS / us / 1e3 / HBM_PEAK_GBS}
        for _, c, plan in lanes:
            lib.qrk_bd_plan_destroy(plan)
    return {"what": "configs[1] as a stream of independent matrices through 1 / 2 / 3 handles on their own HIP streams (wall clock, "
                    "qrk_bd_factorize called from Python per step; 1 = the headline's schedule); not `value`", "handles": res}


def solve_leg(ctx, dev, torch, np):
    """solve() of the block-diagonal solver (_solve_impl, BlockDiagonalSparseQR.h:257-280 -- Q^T b, per-tile back
    substitution with the packed R, the column permutation) for one right-hand side resident in HBM: configs[1]'s matrix, the reference's
    own 7 x 2 blocks and configs[3]'s 8 x 6 left stage; us per solve and the fraction of the HBM roofline at Q, R, b read once, x written."""
    import qrkit_amd
    out = []
    for r, c, B in ((32, 32, BLOCKS), (7, 2, 1000000), (8, 6, 20000), (8, 6, 1000000)):
        g = torch.Generator(device=dev); g.manual_seed(11 * r + c)
        tiles = torch.rand(B * r * c, generator=g, device=dev, dtype=torch.float64) * 4.5 + 0.5
        rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
        qr = qrkit_amd.BlockDiagonalSparseQR(context=ctx)
        qr.compute(qrkit_amd.SparseBlockDiagonal.fromTiles(rows, cols, tiles))
        b = torch.rand(B * r, generator=g, device=dev, dtype=torch.float64)
        for _ in range(3):
            qr.solve(b)
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(10):
                qr.solve(b)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 10)
        by = B * (8 * r * r + 4 * c * (c + 1) + 8 * r + 8 * c + 4 * c)
        out.append({"tile": f"{r}x{c}", "tiles": B, "us": best * 1e6, "algorithmic_GBs": by / best / 1e9, "frac_of_hbm": by / best / 1e9 / HBM_PEAK_GBS})
        del qr, tiles, b
    return out


def small_tiles(ctx, dev, torch, np):
    """Uniform batches of the small and middle tile classes on the final build (tiles/s and the fraction of the HBM roofline at the
    algorithmic bytes of SURVEY.md 8(d): 8 r c in, 8 r^2 + 4 c (c + 1) + 4 c out): 8 x 6 is the left stage of BASELINE configs[3] (20 000 tiles
    there), 7 x 2 and 9 x 2 the reference's own test blocks, 16 x 16 the top of bdqr_quad, 33 x 33 / 64 x 64 the ends of bdqr_w64.  Timed like the
    headline: HIP events around back-to-back factorisations on the handle's stream (qrk_bd_time_factorize), rotating over several
    matrices -- the Python mirror's factorize() allocates three tensors per call and is host-bound below ~ 20 us per launch (it was the
    clock of this leg in round 5: `ms_python_mirror` keeps that figure for the BASELINE-sized batches)."""
    import qrkit_amd
    from qrkit_amd import _capi as capi
    out = []
    for r, c, B in ((7, 2, 1000000), (9, 2, 1000000), (8, 6, 20000), (8, 6, 1000000), (16, 16, 400000), (33, 33, 2000), (33, 33, 20000),
                    (64, 64, 2000), (64, 64, 40000)):
        S = max(1, min(8, 1600000 // (B * r)))
        g = torch.Generator(device=dev); g.manual_seed(7 * r + c)
        tiles = torch.rand(S * B * r * c, generator=g, device=dev, dtype=torch.float64) * 2.0 - 1.0
        q = torch.empty(S * B * r * r, device=dev, dtype=torch.float64)
        rv = torch.empty(S * B * (c * (c + 1) // 2), device=dev, dtype=torch.float64)
        pm = torch.empty(S * B * c, device=dev, dtype=torch.int32)
        lay = capi.BDLayout()
        lay.num_blocks, lay.block_rows, lay.block_cols = B, r, c
        lay.rows = lay.cols = None
        lay.mat_rows, lay.mat_cols = B * r, B * c
        plan = C.c_void_p()
        capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), capi.FULL_Q, capi.COLPIV_HOUSEHOLDER, C.byref(plan)), ctx.handle)
        ms = C.c_float()

        def run(it):
            capi.check(capi.lib().qrk_bd_time_factorize(plan, tiles.data_ptr(), q.data_ptr(), rv.data_ptr(), pm.data_ptr(), S, it, C.byref(ms)),
                       ctx.handle)
            return ms.value * 1e-3
        iters = 50 if B * r * r < 4e7 else 10
        run(max(3, iters // 5))
        best = min(run(iters) for _ in range(3))
        by = 8 * r * c + 8 * r * r + 4 * c * (c + 1) + 4 * c
        e = {"tile": f"{r}x{c}", "tiles": B, "ms": best * 1e3, "tiles_per_s": B / best, "algorithmic_GBs": B * by / best / 1e9,
             "frac_of_hbm": B * by / best / 1e9 / HBM_PEAK_GBS}
        capi.lib().qrk_bd_plan_destroy(plan)
        if B <= 20000:
            rows, cols = np.full(B, r, np.int32), np.full(B, c, np.int32)
            mat = qrkit_amd.SparseBlockDiagonal.fromTiles(rows, cols, tiles[:B * r * c])
            qr = qrkit_amd.BlockDiagonalSparseQR(context=ctx)
            qr.analyzePattern(mat)
            for _ in range(3):
                qr.factorize(mat)
            torch.cuda.synchronize()
            bp = 1e30
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(10):
                    qr.factorize(mat)
                torch.cuda.synchronize()
                bp = min(bp, (time.perf_counter() - t0) / 10)
            e["ms_python_mirror"] = bp * 1e3
            del qr, mat
        out.append(e)
        del tiles, q, rv, pm
    return out


def mixed_share(ctx, dev, torch, np, B=12500):
    """BASELINE configs[4] is 100 000 mixed square tiles, n ~ U{8..256}, over 8 GPUs: one GPU's share, 12 500 tiles (seed 12345), inputs
    resident in HBM, three warm-up factorisations (clocks), best and mean of five timed ones."""
    from qrkit_amd import _capi as capi
    n = np.random.default_rng(12345).integers(8, 257, B).astype(np.int32)
    n64 = n.astype(np.int64)
    lay = capi.BDLayout()
    lay.num_blocks, lay.block_rows, lay.block_cols = B, 0, 0
    lay.rows = n.ctypes.data_as(C.POINTER(C.c_int32)); lay.cols = n.ctypes.data_as(C.POINTER(C.c_int32))
    lay.mat_rows = lay.mat_cols = int(n64.sum())
    plan = C.c_void_p()
    capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), capi.FULL_Q, capi.COLPIV_HOUSEHOLDER, C.byref(plan)), ctx.handle)
    g = torch.Generator(device=dev); g.manual_seed(777)
    t = torch.rand(int((n64 * n64).sum()), generator=g, device=dev, dtype=torch.float64) * 2 - 1
    q = torch.empty(int((n64 * n64).sum()), device=dev, dtype=torch.float64)
    r = torch.empty(int((n64 * (n64 + 1) // 2).sum()), device=dev, dtype=torch.float64)
    p = torch.empty(int(n64.sum()), device=dev, dtype=torch.int32)

    def once():
        capi.check(capi.lib().qrk_bd_factorize(plan, t.data_ptr(), q.data_ptr(), r.data_ptr(), p.data_ptr(), None, capi.MEM_DEVICE), ctx.handle)
        torch.cuda.synchronize()
    for _ in range(3):
        once()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); once(); ts.append(time.perf_counter() - t0)
    capi.lib().qrk_bd_plan_destroy(plan)
    byts = float((8 * n64 * n64 + 8 * n64 * n64 + 4 * n64 * (n64 + 1) + 4 * n64).sum())
    flops = float((2 * n64 ** 3 - 2 * n64 ** 3 / 3 + 4 * (n64 ** 3 / 3)).sum())
    best, mean = min(ts), sum(ts) / len(ts)
    return {"workload": f"{B} square tiles, n ~ U{{8..256}}, seed 12345: the share of one of 8 GPUs of BASELINE configs[4]; ColPivHouseholderQR, explicit Q",
            "ms_best": best * 1e3, "ms_mean": mean * 1e3, "tiles_per_s": B / best, "algorithmic_GBs": byts / best / 1e9, "gflops": flops / best / 1e9}


def strips_config2(ctx, dev, torch, np, N=2048):
    """BASELINE configs[2] (block-banded, 64 x 64 blocks, bandwidth 3): strips of 256 x 192 at column step 64; N of the
    50 000 strips (the chain's time is linear in N; the full size runs in tests/test_banded_strips_gpu.py).  Stage A (every strip
    triangularised, all CUs) is also timed on its own through the block-diagonal solver it is; the chain is the difference."""
    import qrkit_amd
    from qrkit_amd.banded import BandedStripsQR
    ms, n, s = 256, 192, 64
    g = torch.Generator(device=dev); g.manual_seed(5)
    strips = torch.rand(N * ms * n, device=dev, dtype=torch.float64, generator=g) * 2 - 1
    qr = BandedStripsQR(N, ms, n, s, context=ctx)

    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return min(ts)
    t_all = timed(lambda: qr.factorize(strips))
    rows, cols = np.full(N, ms, np.int32), np.full(N, n, np.int32)
    from qrkit_amd import _capi as capi
    bd = qrkit_amd.BlockDiagonalSparseQR(blockSolver=capi.HOUSEHOLDER, qFormat=capi.BLOCK_DIAGONAL_Q, context=ctx, hCoeffs=False)
    mat = qrkit_amd.SparseBlockDiagonal.fromTiles(rows, cols, strips)
    bd.analyzePattern(mat)
    t_a = timed(lambda: bd.factorize(mat))
    b = torch.rand(qr.rows(), device=dev, dtype=torch.float64, generator=g)
    torch.cuda.synchronize(); t0 = time.perf_counter(); qr.solve(b); torch.cuda.synchronize()
    t_solve_first = time.perf_counter() - t0                             # (builds the per-strip maps of banded_maps.hip: once per factorisation)
    t_solve = timed(lambda: qr.solve(b), 2)
    fl_a = N * (2.0 * ms * n * n - 2.0 * n ** 3 / 3)                     # Householder QR of a 256 x 192 strip
    fl_b = N * 5.0e6                                                     # merge of two triangles (staircase), DESIGN.md K4
    return {"workload": f"{N} strips of {ms} x {n}, column step {s} (BASELINE configs[2] has 50 000): qrk_bbs_factorize = stage A (strips "
                        "triangularised as one batch) + stage B (the chain that merges the carried triangle with each strip's)",
            "ms_total": t_all * 1e3, "ms_per_strip": t_all * 1e3 / N, "stage_a_ms": t_a * 1e3, "chain_ms_per_strip": (t_all - t_a) * 1e3 / N,
            "projected_s_for_50000_strips": t_all / N * 50000, "solve_ms_per_strip": t_solve * 1e3 / N,
            "solve_ms": t_solve * 1e3, "solve_first_call_ms": t_solve_first * 1e3,
            "solve_note": "Q^T b and R^-1 y through one small matrix per strip and two-level chains (banded_maps.hip); the first solve after "
                          "a factorisation also builds those matrices; rounds 3-4 (one workgroup walking the strips): 0.071 ms per strip",
            "roofline": {"bound": "fp64 of ONE CU for the chain (a true dependency strip to strip), fp64 of the chip for stage A",
                         "stage_a_TFLOPs": fl_a / t_a / 1e12, "stage_a_frac_of_fp64_peak": fl_a / t_a / 1e12 / 78.6,
                         "chain_GFLOPs": fl_b / max(t_all - t_a, 1e-9) / 1e9, "one_cu_fp64_peak_GFLOPs": 307.0,
                         "chain_frac_of_one_cu": fl_b / max(t_all - t_a, 1e-9) / 1e9 / 307.0}}


def angular_config3(ctx, dev, torch, np, B=20000, r=8, c=6, m2=2000):
    """BASELINE configs[3] on ONE GPU: B tiles of 8 x 6 on the diagonal + a dense right block of (8 B) x 2000 (BlockAngularSparseQR:
    left factor, Q1^T J2, pivoted QR of the bottom rows, makeR); compute() and solve()."""
    import qrkit_amd
    g = torch.Generator(device=dev); g.manual_seed(778)
    tl = torch.rand(B * r * c, generator=g, device=dev, dtype=torch.float64) * 4.5 + 0.5
    left = qrkit_amd.SparseBlockDiagonal.fromTiles(np.full(B, r, np.int32), np.full(B, c, np.int32), tl)
    J2 = (torch.rand(m2, B * r, generator=g, device=dev, dtype=torch.float64) * 4.5 + 0.5).t()       # column-major (B r) x m2
    ba = qrkit_amd.BlockAngularSparseQR(context=ctx)
    mat = qrkit_amd.BlockMatrix1x2(left, J2)
    b = torch.rand(B * r, generator=g, device=dev, dtype=torch.float64)

    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return min(ts)
    t_c = timed(lambda: ba.compute(mat))
    t_s = timed(lambda: ba.solve(b))
    n1, nb = B * r, B * r - B * c
    fl_right = 2.0 * nb * m2 * m2 - 2.0 * m2 ** 3 / 3
    flops = 4.0 * n1 * c * m2 + fl_right
    return {"workload": f"{B} tiles of {r} x {c} + dense {n1} x {m2} right block on one GPU (BASELINE configs[3] un-sharded)",
            "compute_ms": t_c * 1e3, "solve_ms": t_s * 1e3,
            "roofline": {"bound": "mfma (pivoted QR of the dense bottom block, K3), hbm (Q1^T J2)", "TFLOPs": flops / t_c / 1e12,
                         "fp64_peak_TFLOPs": 78.6, "frac_of_fp64_peak": flops / t_c / 1e12 / 78.6,
                         "right_block_flop": fl_right, "J2_bytes": 8.0 * n1 * m2}}


def strong_mixed(args, ctx, dev, dist, backend, rank, world, torch, np):
    """BASELINE configs[4]: B mixed square tiles, n ~ U{8..256} (seed 12345), cut into contiguous ranges balanced by r c^2
    (sharding.shard_ranges); every rank factorises its range; the packed R (about 9 GB at B = 100000) and the permutation are gathered
    on rank 0 with their true byte counts (grouped send/recv) inside the timed region."""
    from qrkit_amd import _capi as capi
    from qrkit_amd.sharding import gather_ragged_to_root, shard_ranges
    B = args.mixed_blocks
    n = np.random.default_rng(12345).integers(8, 257, B).astype(np.int32)
    ranges = shard_ranges(n, n, world)
    lo, hi = ranges[rank]
    nl = np.ascontiguousarray(n[lo:hi])
    n64 = n.astype(np.int64)
    lay = capi.BDLayout()
    lay.num_blocks, lay.block_rows, lay.block_cols = len(nl), 0, 0
    lay.rows = nl.ctypes.data_as(C.POINTER(C.c_int32)); lay.cols = nl.ctypes.data_as(C.POINTER(C.c_int32))
    lay.mat_rows = lay.mat_cols = int(nl.sum())
    plan = C.c_void_p()
    capi.check(capi.lib().qrk_bd_plan_create(ctx.handle, C.byref(lay), capi.FULL_Q, capi.COLPIV_HOUSEHOLDER, C.byref(plan)), ctx.handle)
    l64 = nl.astype(np.int64)
    n_in, n_q, n_r, n_c = int((l64 * l64).sum()), int((l64 * l64).sum()), int((l64 * (l64 + 1) // 2).sum()), int(l64.sum())
    g = torch.Generator(device=dev); g.manual_seed(4242 + rank)
    t = torch.rand(n_in, generator=g, device=dev, dtype=torch.float64) * 2 - 1
    q = torch.empty(n_q, device=dev, dtype=torch.float64)
    r = torch.empty(n_r, device=dev, dtype=torch.float64)
    p = torch.empty(n_c, device=dev, dtype=torch.int32)
    r_sizes = [int((n64[a:b] * (n64[a:b] + 1) // 2).sum()) for a, b in ranges]
    p_sizes = [int(n64[a:b].sum()) for a, b in ranges]
    xdev = dev if backend == "nccl" else "cpu"
    R_all = torch.empty(sum(r_sizes), device=xdev, dtype=torch.float64) if rank == 0 else None
    P_all = torch.empty(sum(p_sizes), device=xdev, dtype=torch.int32) if rank == 0 else None
    base_col = int(n64[:lo].sum())

    def step(gather):
        capi.check(capi.lib().qrk_bd_factorize(plan, t.data_ptr(), q.data_ptr(), r.data_ptr(), p.data_ptr(), None, capi.MEM_DEVICE), ctx.handle)
        if gather:
            gather_ragged_to_root(r.to(xdev), r_sizes, R_all, 0, rank, world)
            gather_ragged_to_root((p + base_col).to(xdev), p_sizes, P_all, 0, rank, world)

    step(True)
    with_g = _timed(dist, torch, dev, backend, lambda: step(True), 3)
    without = _timed(dist, torch, dev, backend, lambda: step(False), 3)
    byts = float((8 * n64 * n64 + 8 * n64 * n64 + 4 * n64 * (n64 + 1) + 4 * n64).sum())
    flops = float((2 * n64 ** 3 - 2 * n64 ** 3 / 3 + 4 * (n64 ** 3 / 3)).sum())
    capi.lib().qrk_bd_plan_destroy(plan)
    return {"workload": f"{B} square tiles, n ~ U{{8..256}}, seed 12345 (BASELINE configs[4]); shards balanced by r c^2",
            "tiles_per_rank": [b - a for a, b in ranges], "ms_with_gather": with_g * 1e3, "ms_factorize_only": without * 1e3,
            "gather_ms": (with_g - without) * 1e3, "tiles_per_s": B / with_g, "gathered_bytes_on_root": 8 * sum(r_sizes) + 4 * sum(p_sizes),
            "roofline": {"bound": "hbm below n ~ 75, fp64 above", "algorithmic_GBs": byts / without / 1e9, "hbm_peak_GBs": HBM_PEAK_GBS * world,
                         "TFLOPs": flops / without / 1e12, "fp64_peak_TFLOPs": 78.6 * world,
                         "frac_of_fp64_peak": flops / without / 1e12 / (78.6 * world)}}


def strong_angular(args, ctx, dev, dist, backend, rank, world, torch, np):
    """BASELINE configs[3]: B tiles of 8 x 6 on the diagonal + a dense right block of (8 B) x 2000, rows sharded with the tiles
    (qrkit_amd.sharding.ShardedBlockAngularQR: left factor and Q1^T J2 on the rank, the bottom rows reduced to one n x n triangle per
    rank by the un-pivoted CAQR, triangles gathered on the root, pivoted right solver there, P2 broadcast); compute() and solve()."""
    import qrkit_amd
    from qrkit_amd.sharding import ShardedBlockAngularQR
    B, r, c, m2 = args.angular_blocks, 8, 6, 2000
    if B * r < 4 * m2:
        m2 = max(16, (B * r // 4) // 16 * 16)
    rows = np.full(B, r, np.int32); cols = np.full(B, c, np.int32)
    slv = ShardedBlockAngularQR(rows, cols, m2, rank, world, context=ctx)
    lo, hi = slv.start, slv.end
    nb = hi - lo
    g = torch.Generator(device=dev); g.manual_seed(777 + rank)
    tl = torch.rand(nb * r * c, generator=g, device=dev, dtype=torch.float64) * 4.5 + 0.5
    left = qrkit_amd.SparseBlockDiagonal.fromTiles(rows[lo:hi], cols[lo:hi], tl)
    J2 = (torch.rand(m2, nb * r, generator=g, device=dev, dtype=torch.float64) * 4.5 + 0.5).t()      # column-major (nb r) x m2
    b = torch.rand(nb * r, generator=g, device=dev, dtype=torch.float64)
    slv.compute(left, J2); slv.solve(b)
    t_compute = _timed(dist, torch, dev, backend, lambda: slv.compute(left, J2), 3)
    t_solve = _timed(dist, torch, dev, backend, lambda: slv.solve(b), 3)
    n1 = B * r
    flops = 4.0 * n1 * c * m2 + (2.0 * (n1 - B * c) * m2 * m2 - 2.0 * m2 ** 3 / 3)          # Q1^T J2 + the right block's QR
    return {"workload": f"{B} tiles of {r}x{c} + dense {n1} x {m2} right block, rows sharded with the tiles (BASELINE configs[3])",
            "tiles_per_rank": [b2 - a for a, b2 in slv.ranges], "compute_ms": t_compute * 1e3, "solve_ms": t_solve * 1e3,
            "factorizations_per_s": 1.0 / t_compute,
            "exchange": f"{world} triangles of {m2} x {m2} doubles gathered on rank 0 ({8 * m2 * m2 * world} bytes), P2 ({4 * m2} bytes) broadcast; solve: one {m2}-vector per rank up, z2 down",
            "roofline": {"bound": "mfma (right block), hbm (Q1^T J2)", "TFLOPs": flops / t_compute / 1e12, "fp64_peak_TFLOPs": 78.6 * world,
                         "frac_of_fp64_peak": flops / t_compute / 1e12 / (78.6 * world), "J2_bytes": 8.0 * n1 * m2,
                         "J2_GBs_if_read_once": 8.0 * n1 * m2 / t_compute / 1e9}}


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
