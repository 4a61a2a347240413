#!/usr/bin/env python3
"""bench.py -- iLQR iterations/sec of the batched hot path on N x MI355X.

    python bench.py --gpus N --steps K --warmup W [--dtype f64|f32] [--batch B]

A "step" is one iLQR iteration over the whole batch: linearise every (b, t), backward
Riccati sweep, rollouts of every trial alpha of the backtracking line search, accept.
Workload (BASELINE.json configs[2], "c3"): under-actuated double pendulum swing-up,
n=4 m=1, N=200, rk4, batch 4096 trajectories PER GPU, all 10 backtracking trials of the
reference (alpha = 1, 1/2, ..., 2^-9) rolled out in parallel in ONE pass -- the config's 8
plus the 2 it would run in a second pass: a second dependent pass costs a full 200-step
sweep latency whatever its width, so one wider pass is strictly cheaper on this chip
(--n-alpha 8 reproduces the two-pass form).  Parameters of run_iLQR_UA_MPC.py:17-67, seeded
random initial states.  Throughput mode (ILQR_FLAG_KEEP_ITERATING): no trajectory ever leaves the
loop, so every step does the full batch's work.  Inputs are resident in HBM before the
timed region.  For N > 1 launch with torch.distributed.run (one rank per GPU, RCCL): each
rank owns an independent shard (weak scaling, no data-path collective); the only exchange
is the per-step all-reduce of {min cost, max |dcost|, #active, #converged}.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the
backward-sweep kernel (HIP events on the kernel's own stream) and `cpu_baseline` (the C
restatement of the reference algorithm, oracle/c, timed on the host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


PHASE_KERNEL = {"linearize": "linearize_kernel", "backward": "backward_tile16_kernel", "forward": "forward_ring_kernel"}


def pmc_traffic(dtype, batch, horizon):
    """HBM bytes per launch of the three hot kernels from the committed rocprofv3 PMC passes (newest
    profiles/rNN/pmc_traffic_*.json for the same dtype, batch, horizon; FETCH_SIZE doubled per the gfx950 correction,
    + WRITE_SIZE).  Counters cannot be read from inside this process, so the figure is the profiled one, else None."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"pmc_traffic_{dtype}.json"))):
        try:
            d = json.load(open(path))
            if d.get("batch") == batch and d.get("horizon") == horizon:
                best = ({ph: float(d["kernels"][k]["hbm_bytes_per_launch"]) for ph, k in PHASE_KERNEL.items()
                         if k in d["kernels"]}, os.path.relpath(path, ROOT))
        except Exception:
            pass
    return best


def host_cores():
    """(share, all): the cores of a 1-GPU job's share of the host (the pool gives one GPU 16 of them:
    ILQR_BENCH_CORES), and every core this process may be scheduled on (its affinity mask)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    n = max(1, min(n, os.cpu_count() or n))
    return max(1, min(n, int(os.environ.get("ILQR_BENCH_CORES", "16")))), n


def numpy_restatement_rate(p, dtype):
    """The NumPy restatement (oracle/ilqr.py: explicit per-timestep loops, the form SURVEY 8d names) on ONE core:
    one trajectory, two iterations of the same problem -- it is ~two orders slower than the C port."""
    from oracle import iLQROracle
    from oracle.build import oracle_from_spec
    from ilqr_amd import problems
    np_dt = np.float64 if dtype == "f64" else np.float32
    x0, U0 = problems.ua_batch(1, seed=0, restarts=False, N=p["N"])
    o = iLQROracle(oracle_from_spec(p["dynamics"], p["cost"], dtype=np_dt), N=p["N"], x_0=x0[0], U_init=U0[0], tol=0.0, maxiter=2)
    t0 = time.perf_counter()
    o.optimize_trajectory()
    return o.iterations / (time.perf_counter() - t0)


def cpu_baseline(p, dtype, cores, budget_s=10.0, total_core_seconds=None):
    """The oracle's C restatement of the reference loop (oracle/c/ilqr_oracle.c: per-timestep backward
    and forward passes, SEQUENTIAL backtracking that stops at the first accepted alpha, one trajectory
    per call) on the host cores: a bounded sample of the same workload, one forked worker per core."""
    import multiprocessing as mp
    from oracle.c_oracle import COracle, build
    from ilqr_amd import problems
    build()
    iters = 10
    np_dt = np.float64 if dtype == "f64" else np.float32
    x0, U0 = problems.ua_batch(256, seed=0, restarts=False, N=p["N"])
    co = COracle(p["dynamics"], p["cost"], dtype=np_dt)
    co.solve(x0[0], U0[0], fixed_iters=iters)  # warm
    t0 = time.perf_counter()
    for i in range(8):
        co.solve(x0[i], U0[i], fixed_iters=iters)
    t1 = (time.perf_counter() - t0) / 8
    if total_core_seconds is not None:     # fixed amount of WORK, spread over `cores` workers (see main)
        per_core = max(2, int(total_core_seconds / max(t1, 1e-6) / cores))
    else:
        per_core = max(8, int(budget_s / max(t1, 1e-6)))
    n_traj = per_core * cores

    def work(rank, q):
        c = COracle(p["dynamics"], p["cost"], dtype=np_dt)
        for i in range(rank * per_core, (rank + 1) * per_core):
            c.solve(x0[i % len(x0)], U0[i % len(x0)], fixed_iters=iters)
        q.put(rank)

    ctx = mp.get_context("fork")
    q = ctx.Queue()
    procs = [ctx.Process(target=work, args=(r, q)) for r in range(cores)]
    t0 = time.perf_counter()
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join()
    wall = time.perf_counter() - t0
    return {"value": n_traj * iters / wall, "unit": "iLQR iterations/sec", "cores": cores, "kind": "port",
            "single_core_value": iters / t1, "host_cpu_count": os.cpu_count(), "affinity_cores": host_cores()[1],
            "sample": f"{n_traj} trajectories x {iters} iterations of the same c3 problem ({dtype}); C restatement of the "
                      f"reference loop with its sequential backtracking; {cores} worker processes, {wall:.1f} s"}


def mpc_c4_extra(ilqr_amd, _lib, problems, np_dt, device, stream, B=1024, n_sim=10):
    """BASELINE config c4 at one GPU's shard, reported beside the headline (outside every timed region above): 1024
    warm-started MPC instances of the under-actuated double pendulum, N = 200, rk4 optimiser, backward_euler plant,
    tol 1e-5, maxiter 50 (run_iLQR_UA_MPC.py:17-174), `n_sim` receding-horizon steps device-resident (ilqr_mpc_run)."""
    p = problems.ua_double_pendulum(N=200)
    x0, U0 = problems.ua_batch(B, seed=2, restarts=False, N=200)
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], np_dt)
    h = sysm.make_handle(horizon=200, batch=B, n_alpha=10, n_trials=10, tol=p["tol"], maxiter=p["maxiter"],
                         plant_integrator="backward_euler", device=device, stream=stream)
    h.mpc_reset(x0, U0)
    h.mpc_run(2)                       # cold start: the first solves run to maxiter; not part of the steady figure
    t0 = time.perf_counter()
    u, x, c = h.mpc_run(n_sim)         # returns after the logs have been copied back (synchronous)
    wall = time.perf_counter() - t0
    its = h.get(_lib.ITERS)
    h.close()
    return {"instances": B, "horizon": 200, "maxiter": p["maxiter"], "steps": n_sim, "ms_per_mpc_step": 1e3 * wall / n_sim,
            "instance_steps_per_sec": B * n_sim / wall, "iterations_last_step_mean": float(np.mean(its)),
            "iterations_last_step_max": int(np.max(its)), "all_finite": bool(np.isfinite(c).all())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="trajectories per GPU")
    ap.add_argument("--n-alpha", type=int, default=10)
    ap.add_argument("--dtype", default="f32", choices=["f64", "f32"],
                    help="f32 = the reference's own (JAX default) precision; f64 = the build's double mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-phase-timing", action="store_true")
    ap.add_argument("--no-solve-extra", action="store_true", help="skip the solve_to_convergence / MPC extras (profiling runs)")
    ap.add_argument("--exchange", action="store_true",
                    help="N = 1 only: run the inter-GPU status exchange anyway, over a ONE-rank RCCL group")
    ap.add_argument("--exchange-every", type=int, default=0,
                    help="iterations between two status exchanges (N > 1 or --exchange); 0 = once, behind the last of the "
                         "K steps -- a solve needs the global status once, at its end, and no shard needs another shard's "
                         "numbers to iterate.  One exchange costs the compute stream 25-50 us (status reduce, event, and a "
                         "collective kernel sharing the CUs): measured with --exchange --exchange-every 1 / 4")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import ilqr_amd
    from ilqr_amd import _lib, problems
    p = problems.ua_double_pendulum(integrator="rk4", N=200)
    # CPU baseline first: its worker processes are forked before this process touches the GPU
    cpu = cpu_all = None
    if world == 1 and not args.no_cpu_baseline:
        share, every = host_cores()
        cpu = cpu_baseline(p, args.dtype, share)
        cpu["numpy_restatement_single_core_value"] = numpy_restatement_rate(p, args.dtype)
        if every > share:
            # P = every core the job may be scheduled on.  The pool gives a 1-GPU job the CPU time of its
            # share whatever the affinity mask says (round 2: 256 workers ran at 0.44x the 16-worker rate), so this leg
            # gets a fixed amount of WORK -- as many core-seconds as the share leg -- not a fixed time per worker: a
            # fraction of a second if the cores are really there, about as long as the share leg if they are not.
            cpu_all = cpu_baseline(p, args.dtype, every, total_core_seconds=10.0 * share)
            cpu_all["note"] = ("every core in the affinity mask; the pool throttles a 1-GPU job to its share of the host, so "
                               "this figure says what the extra workers bought, not what the whole host could do")

    import torch
    import torch.distributed as dist
    from ilqr_amd.dist import StatusExchange
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus}` "
                         f"(WORLD_SIZE is {world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the measured path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")  # RCCL
    elif args.exchange:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]))
        dist.init_process_group("nccl", rank=0, world_size=1)
    exchange = world > 1 or args.exchange

    np_dt = np.float64 if args.dtype == "f64" else np.float32
    B, N = args.batch, p["N"]
    x0, U0 = problems.ua_batch(B, seed=1000 + rank, restarts=False, N=N)
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], np_dt)
    # launch on an explicit torch stream (made current) so torch's barriers, synchronize() and events see
    # the kernels: torch.cuda.Event only observes torch's current stream, and a NULL stream pointer would
    # make the handle create a private one
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream, "expected a non-null HIP stream from torch"
    h = sysm.make_handle(horizon=N, batch=B, n_alpha=args.n_alpha, n_trials=10, tol=p["tol"], maxiter=1 << 30,
                         device=local_rank, flags=_lib.FLAG_KEEP_ITERATING, stream=stream)
    h.set_problem(x0, U0)       # uploads: inputs are HBM-resident from here on
    h.initial_rollout()
    xchg = StatusExchange(device=f"cuda:{local_rank}") if exchange else None

    every = args.exchange_every if args.exchange_every > 0 else args.steps
    count = [0]

    def step():
        h.iterate(1)
        count[0] += 1
        if exchange and count[0] % every == 0:
            # the path's only inter-GPU exchange: best cost / convergence, as one 32-B all-gather
            # over RCCL on a side stream, so the compute stream never waits for it
            xchg.launch(lambda t: h.status_reduce(t.data_ptr()))

    def fence():
        if exchange and xchg.k:
            xchg.result()            # the last exchange has landed on every rank
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if exchange:
        # the collective's first call builds the RCCL communicator (tens of ms): it belongs to the warm-up
        xchg.launch(lambda t: h.status_reduce(t.data_ptr()))
    count[0] = 0
    fence()
    # ---- the timed region: exactly K steps, nothing else on the stream ------------------------------
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    wall = time.perf_counter() - t0

    # ---- per-phase breakdown: the same K steps again with HIP start/stop events attached to every kernel
    # dispatch (hipExtLaunchKernelGGL: the dispatch's own begin/end timestamps, no extra stream packets)
    phases = None
    bwd_us = None
    if not args.no_phase_timing:
        h.timing_enable(True)
        h.timing_reset()
        for _ in range(args.steps):
            h.iterate(1)
        phases = h.timing_get()
        h.timing_enable(False)
        # ---- extra: R back-to-back launches of the backward sweep on the same expansion (idempotent: reads
        # lin/term, rewrites the same gains) between ONE pair of torch events on this stream.  It runs
        # faster than inside the iteration (its tiles are then still cached from the previous launch instead
        # of being freshly written by linearise) and is reported separately, not used for the roofline.
        R = 50
        h.linearize()
        for _ in range(5):
            h.backward()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(R):
            h.backward()
        e1.record()
        torch.cuda.synchronize()
        bwd_us = e0.elapsed_time(e1) * 1e3 / R

    wall_t = torch.tensor([wall], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall = float(wall_t.item())
    cost = h.get(_lib.COST)
    finite = bool(np.isfinite(cost).all())

    if rank == 0:
        value = world * B * args.steps / wall
        out = {
            "metric": "iLQR iterations/sec (batch=4096, T=200, n=4 m=1); backward-pass HBM GB/s",
            "value": value, "unit": "iLQR iterations/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * wall / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "c3: under-actuated double pendulum swing-up (run_iLQR_UA_MPC.py params), "
                                   f"n=4 m=1 N=200 rk4, batch {B} trajectories per GPU, {args.n_alpha} parallel "
                                   "line-search alphas per pass covering the 10 reference trials, fixed iterations",
                       "batch_per_gpu": B, "horizon": N, "n_alpha": args.n_alpha, "n_trials": 10,
                       "sharding": f"{world} independent shards, scalar status all-gather every {every} steps (once per solve)" if world > 1
                       else "single shard"},
            "all_costs_finite": finite,
        }
        if phases is not None:
            ab = h.algorithmic_bytes()
            ms, n = phases["backward"]
            avg_s = ms / max(n, 1) * 1e-3      # in-iteration launches of the profiled K steps
            achieved = ab["backward"] / avg_s / 1e9
            tr = pmc_traffic(args.dtype, B, N)
            out["roofline"] = {"bound": "hbm", "kernel": "backward Riccati sweep (backward_tile16_kernel)",
                               "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": tr[0].get("backward") if tr else None,
                               "traffic_source": tr[1] if tr else None,
                               "algorithmic_bytes_per_launch": ab["backward"], "avg_launch_us": avg_s * 1e6,
                               "launches": n, "avg_launch_us_back_to_back": bwd_us,
                               "frac_of_measured_copy_peak_6.29TBs": achieved / 6290.0}
            out["phases_us_per_step"] = {k: 1e3 * v[0] / args.steps for k, v in phases.items()}
            # the same accounting for every hot kernel: algorithmic bytes / its average launch
            out["kernels"] = {}
            for ph, kern in PHASE_KERNEL.items():
                ms_k, n_k = phases[ph]
                if n_k:
                    ach = ab[ph] / (ms_k / n_k * 1e-3) / 1e9
                    out["kernels"][kern] = {"avg_launch_us": ms_k / n_k * 1e3, "launches": n_k,
                                            "algorithmic_bytes_per_launch": ab[ph], "achieved_GBs": ach,
                                            "frac_of_8TBs": ach / HBM_PEAK_GBS,
                                            "traffic": tr[0].get(ph) if tr else None}
        if os.environ.get("ILQR_CLOCK_PROBE"):
            pr = h.get(_lib.PROBE)
            out["clock_probe"] = {"backward_cycles": int(pr[0]), "backward_GHz": float(pr[0]) / max(float(pr[1]), 1) * 0.1,
                                  "forward_cycles": int(pr[2]), "forward_GHz": float(pr[2]) / max(float(pr[3]), 1) * 0.1}
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["gpu_over_cpu"] = value / cpu["value"]
            if cpu_all is not None:
                out["cpu_baseline_all_cores"] = cpu_all
                out["gpu_over_cpu_all_cores"] = value / cpu_all["value"]
        if exchange and world == 1:
            out["exchange"] = f"status all-gather over a one-rank RCCL group (side stream) every {every} steps"
        if world == 1 and not args.no_phase_timing and not args.no_solve_extra:
            # reported beside the throughput figure, outside every timed region above: the same batch
            # solved to convergence with the reference's stopping rules (tol, maxiter 50, line-search failure)
            hs = sysm.make_handle(horizon=N, batch=B, n_alpha=args.n_alpha, n_trials=10, tol=p["tol"], maxiter=50,
                                  device=local_rank, stream=stream)
            hs.set_problem(x0, U0)
            hs.solve()                       # warm-up (kernel paging, event creation)
            hs.set_problem(x0, U0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            its, _ = hs.solve()
            t_solve = time.perf_counter() - t0
            stw = hs.get(_lib.STATUS) & 0xff
            out["solve_to_convergence"] = {
                "batch": B, "wall_ms": t_solve * 1e3, "iterations_mean": float(np.mean(its)), "iterations_max": int(np.max(its)),
                "converged": int(np.sum(stw == _lib.TRAJ_CONVERGED)), "linesearch_failed": int(np.sum(stw == _lib.TRAJ_LINESEARCH_FAILED)),
                "maxiter": int(np.sum(stw == _lib.TRAJ_MAXITER)), "tol": p["tol"]}
            hs.close()
            out["mpc_c4_shard"] = mpc_c4_extra(ilqr_amd, _lib, problems, np_dt, local_rank, stream)
        print(json.dumps(out))
    if world > 1 or args.exchange:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
