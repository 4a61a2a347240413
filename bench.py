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


PHASE_KERNEL = {"linearize": "linearize_kernel", "backward": "backward_tile16_kernel", "forward": "forward_ring_kernel",
                "fused": "backward_fused16_kernel"}


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


def rank_barrier(dist, dev):
    """Barrier over the ranks: RCCL needs the device named; gloo (several ranks rehearsed on one card) takes none."""
    if dist.get_backend() == "nccl":
        dist.barrier(device_ids=[dev])
    else:
        dist.barrier()


def max_over_ranks(torch, dist, value, world):
    if world <= 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def kernel_table(phases, ab, traffic, names, peak=HBM_PEAK_GBS):
    """{kernel: {avg_launch_us, launches, algorithmic_bytes_per_launch, achieved_GBs, frac_of_8TBs, traffic}} from the
    handle's per-dispatch HIP events (phases: {phase: (ms, launches)}) and its algorithmic byte counts."""
    out = {}
    for ph, kern in names.items():
        ms_k, n_k = phases.get(ph, (0.0, 0))
        if not n_k:
            continue
        ach = ab[ph] / (ms_k / n_k * 1e-3) / 1e9
        out[kern] = {"avg_launch_us": ms_k / n_k * 1e3, "launches": n_k, "algorithmic_bytes_per_launch": ab[ph],
                     "achieved_GBs": ach, "frac_of_8TBs": ach / peak, "traffic": (traffic or {}).get(ph)}
    return out


def pmc_valu(dtype, batch, horizon):
    """Vector instructions per launch of the hot kernels (SQ_INSTS_VALU, summed over the waves of a launch) from the
    newest committed profiles/rNN/pmc_valu_<dtype>.json of the same shape, or None."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"pmc_valu_{dtype}.json"))):
        try:
            d = json.load(open(path))
            if d.get("batch") == batch and d.get("horizon") == horizon:
                best = ({k: float(v["SQ_INSTS_VALU_per_launch"]) for k, v in d["kernels"].items()}, os.path.relpath(path, ROOT))
        except Exception:
            pass
    return best


def timed_iterations(h, steps, warm=3):
    """(wall seconds per iteration, {phase: (ms, launches)}) of `steps` iterations on a handle whose problem is set."""
    h.initial_rollout()
    h.iterate(warm)
    h.sync()
    t0 = time.perf_counter()
    h.iterate(steps)
    h.sync()                      # (completes the last iteration's acceptance step too)
    wall = (time.perf_counter() - t0) / steps
    h.timing_enable(True)
    h.timing_reset()
    for _ in range(steps):        # (one call per iteration: a launch of the persistent kernel is then one iteration)
        h.iterate(1)
    h.flush()
    ph = h.timing_get()
    h.timing_enable(False)
    return wall, ph


FUSED_NAMES = {"persist": "ilqr_persistent_kernel (one launch = one whole iteration of the batch)",
               "fused": "backward_fused16_kernel", "forward": "forward_ring_kernel", "linearize": "linearize_kernel",
               "backward": "backward_tile16_kernel", "select": "select_kernel"}
C5_NAMES = {"linearize": "linearize_wave_kernel", "backward": "backward_mfma16_kernel", "forward": "forward_mfma16_kernel",
            "select": "select_kernel"}


def config_extras(ilqr_amd, _lib, problems, device, stream, steps=10):
    """The other BASELINE configs on this GPU, outside every timed region of the headline: per-iteration time,
    iterations/sec and the per-kernel roofline table (algorithmic bytes of SURVEY 8d / HIP-event launch time), fixed
    iterations (ILQR_FLAG_KEEP_ITERATING) like the headline.  c1 as the driver runs it (N = 400, backward Euler, B = 1),
    c2 (B = 256), the c5 shard (B = 128) and c5 whole (B = 1024), c3 in fp64."""
    out = {}

    def run(tag, sysm, x0, U0, N, names, what):
        h = sysm.make_handle(horizon=N, batch=len(x0), n_alpha=10, n_trials=10, maxiter=1 << 30, device=device,
                             flags=_lib.FLAG_KEEP_ITERATING, stream=stream)
        h.set_problem(x0, U0)
        wall, ph = timed_iterations(h, steps)
        ab = h.algorithmic_bytes()
        out[tag] = {"workload": what, "batch": len(x0), "horizon": N, "dtype": "f32" if sysm.dtype == np.float32 else "f64",
                    "ms_per_iteration": wall * 1e3, "iterations_per_sec": len(x0) / wall,
                    "phases_us_per_iteration": {k: 1e3 * v[0] / steps for k, v in ph.items() if v[1]},
                    "kernels": kernel_table(ph, ab, None, names)}
        h.close()

    p = problems.pendulum_open_loop(integrator="backward_euler", N=400)
    run("c1", ilqr_amd.make_system(p["dynamics"], p["cost"], np.float32), p["x0"][None], p["U_init"][None], 400, FUSED_NAMES,
        "pendulum n=2 m=1, N=400 backward_euler, batch 1 (run_iLQR_open_loop.py:16-69)")
    p = problems.ua_double_pendulum()
    x0, U0 = problems.ua_batch(256, seed=0)
    run("c2", ilqr_amd.make_system(p["dynamics"], p["cost"], np.float32), x0, U0, 200, FUSED_NAMES,
        "UA double pendulum n=4 m=1, N=200 rk4, batch 256")
    x0, U0 = problems.ua_batch(4096, seed=1000)
    run("c3_f64", ilqr_amd.make_system(p["dynamics"], p["cost"], np.float64), x0, U0, 200, FUSED_NAMES,
        "the headline workload in fp64")
    # (not a BASELINE config: the reference's third system, run_MPC_double_pendulum.py -- n_u = 2 on the same fused path)
    p = problems.double_pendulum(N=100)
    rng = np.random.default_rng(7)
    run("dp_4x2", ilqr_amd.make_system(p["dynamics"], p["cost"], np.float32), np.asarray(p["x0"])[None] + 0.1 * rng.standard_normal((4096, 4)),
        np.zeros((4096, 2, 100)), 100, FUSED_NAMES, "fully actuated double pendulum n=4 m=2, N=100 rk4, batch 4096")
    p = problems.linear_quadratic()
    for B, tag in ((128, "c5_shard"), (1024, "c5")):
        x0, U0 = problems.lq_batch(B, 16, 8, 500)
        run(tag, ilqr_amd.make_system(p["dynamics"], p["cost"], np.float32), x0, U0, 500, C5_NAMES,
            f"synthetic linear-quadratic n=16 m=8, N=500, batch {B}" + (" (one GPU's share of 1024 over 8)" if B == 128 else ""))
    return out


def mpc_extra(ilqr_amd, _lib, problems, np_dt, device, stream, B=1024, n_sim=10):
    """BASELINE config c4 at one GPU's shard, reported beside the headline (outside every timed region above): 1024
    warm-started MPC instances of the under-actuated double pendulum, N = 200, rk4 optimiser, backward_euler plant,
    tol 1e-5, maxiter 50 (run_iLQR_UA_MPC.py:17-174), `n_sim` receding-horizon steps device-resident (ilqr_mpc_run).
    Default path: ONE persistent launch for all steps (every workgroup paces its own instances).  Beside it the
    host-looped form (ILQR_FLAG_NO_PERSIST: a step lasts until the batch's slowest instance has converged) with the
    step's attribution from the per-dispatch HIP events."""
    p = problems.ua_double_pendulum(N=200)
    x0, U0 = problems.ua_batch(B, seed=2, restarts=False, N=200)
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], np_dt)

    def run(flags, timed_phases):
        h = sysm.make_handle(horizon=200, batch=B, n_alpha=10, n_trials=10, tol=p["tol"], maxiter=p["maxiter"],
                             plant_integrator="backward_euler", device=device, stream=stream, flags=flags)
        h.mpc_reset(x0, U0)
        h.mpc_run(2)                       # cold start: the first solves run to maxiter; not part of the steady figure
        t0 = time.perf_counter()
        u, x, c = h.mpc_run(n_sim)         # returns after the logs have been copied back (synchronous)
        wall = time.perf_counter() - t0
        its = h.get(_lib.ITERS)
        ph = wall_t = None
        if timed_phases:
            h.timing_enable(True)
            h.timing_reset()
            t0 = time.perf_counter()
            h.mpc_run(n_sim)
            wall_t = time.perf_counter() - t0
            ph = h.timing_get()
            h.timing_enable(False)
        h.close()
        return wall, its, c, ph, wall_t

    wall, its, c, _, _ = run(0, False)
    out = {"instances": B, "horizon": 200, "maxiter": p["maxiter"], "steps": n_sim, "ms_per_mpc_step": 1e3 * wall / n_sim,
           "instance_steps_per_sec": B * n_sim / wall, "iterations_last_step_mean": float(np.mean(its)),
           "iterations_last_step_max": int(np.max(its)), "all_finite": bool(np.isfinite(c).all()),
           "path": "persistent kernel: one launch for all steps, every workgroup (4 instances) paces its own solves"}
    wall_l, its_l, c_l, ph, wall_t = run(_lib.FLAG_NO_PERSIST, True)
    busy_ms = sum(v[0] for v in ph.values())
    launches = {k: v[1] for k, v in ph.items() if v[1]}
    per_launch = {k: 1e3 * v[0] / v[1] for k, v in ph.items() if v[1]}
    n_iter_launch = max(ph.get("fused", (0, 0))[1], ph.get("backward", (0, 0))[1])
    out["host_looped"] = {
        "ms_per_mpc_step": 1e3 * wall_l / n_sim, "instance_steps_per_sec": B * n_sim / wall_l,
        "same_result": bool(np.array_equal(c, c_l)),
        "attribution": {"iteration_launches_per_step": n_iter_launch / n_sim, "kernel_us_per_launch": per_launch,
                        "launches_per_step": {k: v / n_sim for k, v in launches.items()},
                        "kernel_busy_ms_per_step": busy_ms / n_sim,
                        "host_and_gaps_ms_per_step": 1e3 * wall_t / n_sim - busy_ms / n_sim,
                        "note": "a step lasts until the batch's slowest instance has converged: iteration launches per step x (fused + "
                                "rollout) + one plant step; HIP-event time of every dispatch of a second run of the same steps; "
                                "'other' = the plant step (mpc_advance_kernel)"}}
    return out


def run_c4(args, world, rank, local_rank, ilqr_amd, _lib, problems, torch, dist):
    """BASELINE c4: 8192 warm-started MPC instances of the UA double pendulum (run_iLQR_UA_MPC.py:146-174), sharded
    over the ranks (shard_range: 1024 per GPU at 8), device-resident; a step = one receding-horizon step of every
    instance.  The only exchange is the scalar status all-reduce behind the timed steps (ShardedBatch.global_status)."""
    from ilqr_amd.dist import ShardedBatch
    total = args.batch if args.batch else 8192
    p = problems.ua_double_pendulum(N=200)
    x0, U0 = problems.ua_batch(total, seed=2, restarts=False, N=200)
    np_dt = np.float64 if args.dtype == "f64" else np.float32
    plant = ilqr_amd.make_system(dict(p["dynamics"], integrator=p["plant_integrator"]), p["cost"], np_dt)
    sb = ShardedBatch(lambda: ilqr_amd.make_system(p["dynamics"], p["cost"], np_dt), x0.astype(np_dt), U0.astype(np_dt),
                      device=local_rank, N=200, tol=p["tol"], maxiter=p["maxiter"], plant=plant, n_alpha=10)
    sb.mpc_reset()
    sb.mpc_run(max(args.warmup, 1))       # cold start: the first solves run to maxiter
    if world > 1 or dist.is_initialized():
        rank_barrier(dist, local_rank)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    u, x, c = sb.mpc_run(args.steps)
    st = sb.global_status()
    if world > 1 or dist.is_initialized():
        rank_barrier(dist, local_rank)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    wall = max_over_ranks(torch, dist, wall, world)
    if rank == 0:
        print(json.dumps({
            "metric": "MPC instance-steps/sec (8192 warm-started UA double pendulum instances, T=200, n=4 m=1)",
            "value": total * args.steps / wall, "unit": "MPC instance-steps/sec", "n_gpus": world, "steps": args.steps,
            "warmup": max(args.warmup, 1), "ms_per_step": 1e3 * wall / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "c4: 8192 MPC instances (run_iLQR_UA_MPC.py), rk4 optimiser, backward_euler plant, tol 1e-5, "
                                   "maxiter 50, device-resident receding-horizon loop", "instances": total,
                       "instances_per_gpu": sb.hi - sb.lo, "horizon": 200,
                       "sharding": f"{world} contiguous shards (dist.shard_range), one scalar status all-reduce behind the steps"},
            "global_status": {"min_cost": st.min_cost, "n_active": st.n_active, "n_converged": st.n_converged},
            "all_costs_finite": bool(np.isfinite(c).all())}))


def run_c5(args, world, rank, local_rank, ilqr_amd, _lib, problems, torch, dist):
    """BASELINE c5: synthetic linear-quadratic system n=16 m=8, N=500, 1024 trajectories sharded over the ranks (128
    per GPU at 8); a step = one iLQR iteration of the shard (MFMA sweep + MFMA rollouts), fixed iterations."""
    from ilqr_amd.dist import shard_range
    total = args.batch if args.batch else 1024
    lo, hi = shard_range(total, world, rank)
    p = problems.linear_quadratic()
    x0, U0 = problems.lq_batch(total, 16, 8, 500)
    np_dt = np.float64 if args.dtype == "f64" else np.float32
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    h = ilqr_amd.make_system(p["dynamics"], p["cost"], np_dt).make_handle(
        horizon=500, batch=hi - lo, n_alpha=10, n_trials=10, maxiter=1 << 30, device=local_rank,
        flags=_lib.FLAG_KEEP_ITERATING, stream=tstream.cuda_stream)
    h.set_problem(x0[lo:hi], U0[lo:hi])
    h.initial_rollout()
    h.iterate(args.warmup)
    stats = torch.zeros(4, dtype=torch.float64, device="cuda")
    from ilqr_amd.dist import allreduce_status, to_status
    if world > 1 or dist.is_initialized():
        rank_barrier(dist, local_rank)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h.iterate(args.steps)
    h.status_reduce(stats.data_ptr())
    allreduce_status(stats)
    if world > 1 or dist.is_initialized():
        rank_barrier(dist, local_rank)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    wall = max_over_ranks(torch, dist, wall, world)
    st = to_status(stats.cpu())
    if rank == 0:
        print(json.dumps({
            "metric": "iLQR iterations/sec (synthetic LQ n=16 m=8, T=500, batch=1024)",
            "value": total * args.steps / wall, "unit": "iLQR iterations/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * wall / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "c5: synthetic linear-quadratic system n=16 m=8 N=500, fixed iterations", "batch": total,
                       "batch_per_gpu": hi - lo, "horizon": 500, "n_alpha": 10,
                       "sharding": f"{world} contiguous shards (dist.shard_range), one scalar status all-reduce behind the steps"},
            "global_status": {"min_cost": st.min_cost, "n_active": st.n_active}}))
    h.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c3", choices=["c3", "c4", "c5"],
                    help="c3 = the headline (BASELINE metric); c4 = 8192 MPC instances, c5 = LQ n=16 m=8 batch 1024: both "
                         "sharded over the ranks of torch.distributed.run, lines of their own")
    ap.add_argument("--batch", type=int, default=0, help="c3: trajectories per GPU (4096); c4 / c5: total over all ranks")
    ap.add_argument("--n-alpha", type=int, default=10)
    ap.add_argument("--dtype", default="f32", choices=["f64", "f32"],
                    help="f32 = the reference's own (JAX default) precision; f64 = the build's double mode")
    ap.add_argument("--materialised", action="store_true",
                    help="time the four-launch iteration over the materialised expansion (ILQR_FLAG_NO_FUSE) instead of the fused one")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (one rank per GPU); gloo = rehearsal of the N-rank control flow with several ranks "
                         "on ONE card (tests): ranks map to device LOCAL_RANK % device_count, collectives run on host copies")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-phase-timing", action="store_true")
    ap.add_argument("--no-solve-extra", action="store_true", help="skip the solve_to_convergence / MPC / configs extras (profiling runs)")
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
    if world == 1 and not args.no_cpu_baseline and args.config == "c3":
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
    local_rank = local_rank % torch.cuda.device_count()      # (only differs from LOCAL_RANK in the one-card rehearsal)
    torch.cuda.set_device(local_rank)
    if world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ and (args.exchange or args.config != "c3")):
        # under torch.distributed.run (any world size): rendezvous through the launcher's own store
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend)  # nccl = RCCL
    elif args.exchange or args.config != "c3":
        # one rank: the collectives of the sharded configs still go through RCCL (a group of ONE rank)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]))
        dist.init_process_group("nccl", rank=0, world_size=1)
    if args.config != "c3":
        (run_c4 if args.config == "c4" else run_c5)(args, world, rank, local_rank, ilqr_amd, _lib, problems, torch, dist)
        rank_barrier(dist, local_rank)
        dist.destroy_process_group()
        return
    exchange = world > 1 or args.exchange

    np_dt = np.float64 if args.dtype == "f64" else np.float32
    B, N = (args.batch or 4096), p["N"]
    x0, U0 = problems.ua_batch(B, seed=1000 + rank, restarts=False, N=N)
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], np_dt)
    # launch on an explicit torch stream (made current) so torch's barriers, synchronize() and events see
    # the kernels: torch.cuda.Event only observes torch's current stream, and a NULL stream pointer would
    # make the handle create a private one
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream, "expected a non-null HIP stream from torch"
    path_flags = _lib.FLAG_NO_FUSE if args.materialised else 0
    h = sysm.make_handle(horizon=N, batch=B, n_alpha=args.n_alpha, n_trials=10, tol=p["tol"], maxiter=1 << 30,
                         device=local_rank, flags=_lib.FLAG_KEEP_ITERATING | path_flags, stream=stream)
    h.set_problem(x0, U0)       # uploads: inputs are HBM-resident from here on
    h.initial_rollout()
    xchg = StatusExchange(device=f"cuda:{local_rank}") if exchange else None

    every = args.exchange_every if args.exchange_every > 0 else args.steps
    count = [0]

    def run_steps(n):
        """n steps: the iterations between two exchanges are ONE call into the library (its C++ loop enqueues the two
        launches of every step back to back; a Python call per step left the stream waiting for the host on a slow box:
        9 us of gaps per step measured on one, 2 us on another)."""
        done = 0
        while done < n:
            chunk = min(every - count[0] % every, n - done) if exchange else n - done
            h.iterate(chunk)
            done += chunk
            count[0] += chunk
            if exchange and count[0] % every == 0:
                # the path's only inter-GPU exchange: best cost / convergence, as one 32-B all-gather
                # over RCCL on a side stream, so the compute stream never waits for it
                xchg.launch(lambda t: h.status_reduce(t.data_ptr()))

    def fence():
        h.flush()                    # the acceptance step of the newest candidates, if the last launch left it pending
        if exchange and xchg.k:
            xchg.result()            # the last exchange has landed on every rank
            rank_barrier(dist, local_rank)
        torch.cuda.synchronize()

    run_steps(args.warmup)
    if exchange:
        # the collective's first call builds the RCCL communicator (tens of ms): it belongs to the warm-up
        xchg.launch(lambda t: h.status_reduce(t.data_ptr()))
    count[0] = 0
    fence()
    # ---- the timed region: exactly K steps (each: acceptance of the previous candidates, linearise, sweep, all
    # rollouts), closed by the acceptance step of the last one; nothing else on the stream --------------------------
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    wall = time.perf_counter() - t0

    # ---- per-phase breakdown: the same K steps again with HIP start/stop events attached to every kernel
    # dispatch (hipExtLaunchKernelGGL: the dispatch's own begin/end timestamps, no extra stream packets)
    phases = None
    mat = None
    if not args.no_phase_timing:
        h.timing_enable(True)
        h.timing_reset()
        for _ in range(args.steps):
            h.iterate(1)
        h.flush()
        phases = h.timing_get()
        h.timing_enable(False)
        # ---- the materialised path beside it (outside the timed region): linearize_kernel -> backward_tile16_kernel ->
        # rollouts -> select_kernel over the expansion in HBM, the form SURVEY 8(d)'s byte count describes.  Same
        # problem, same K steps; plus R back-to-back launches of the sweep on one expansion (its tiles are then
        # still cached from the previous launch instead of freshly written by linearise: reported separately).
        hm = h if args.materialised else sysm.make_handle(
            horizon=N, batch=B, n_alpha=args.n_alpha, n_trials=10, tol=p["tol"], maxiter=1 << 30, device=local_rank,
            flags=_lib.FLAG_KEEP_ITERATING | _lib.FLAG_NO_FUSE, stream=stream)
        if hm is not h:
            hm.set_problem(x0, U0)
        wall_m, ph_m = (wall / args.steps, phases) if args.materialised else timed_iterations(hm, args.steps, warm=args.warmup)
        R = 50
        hm.linearize()
        for _ in range(5):
            hm.backward()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(R):
            hm.backward()
        e1.record()
        torch.cuda.synchronize()
        mat = {"wall_s_per_step": wall_m, "phases": ph_m, "bwd_us_back_to_back": e0.elapsed_time(e1) * 1e3 / R,
               "ab": hm.algorithmic_bytes()}
        if hm is not h:
            hm.close()

    wall = max_over_ranks(torch, dist, wall, world)
    cost = h.get(_lib.COST)
    finite = bool(np.isfinite(cost).all())

    if rank == 0:
        value = world * B * args.steps / wall
        path = ("materialised: linearize_kernel, backward_tile16_kernel, forward_ring_kernel, select_kernel (4 launches)"
                if args.materialised else
                "persistent: ilqr_persistent_kernel (the whole iteration of a shard of <= 1024 trajectories in one launch)"
                if B <= int(os.environ.get("ILQR_PERSIST_ITERATE_MAX", "1024")) and args.dtype == "f32" else
                "fused: backward_fused16_kernel (acceptance step + linearise + sweep, tiles through LDS), forward_ring_kernel (2 launches)")
        out = {
            "metric": "iLQR iterations/sec (batch=4096, T=200, n=4 m=1); backward-pass HBM GB/s",
            "value": value, "unit": "iLQR iterations/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * wall / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "c3: under-actuated double pendulum swing-up (run_iLQR_UA_MPC.py params), "
                                   f"n=4 m=1 N=200 rk4, batch {B} trajectories per GPU, {args.n_alpha} parallel "
                                   "line-search alphas per pass covering the 10 reference trials, fixed iterations",
                       "batch_per_gpu": B, "horizon": N, "n_alpha": args.n_alpha, "n_trials": 10, "iteration_path": path,
                       "sharding": f"{world} independent shards, scalar status all-gather every {every} steps (once per solve)" if world > 1
                       else "single shard"},
            "all_costs_finite": finite,
        }
        if phases is not None:
            ab = h.algorithmic_bytes()
            tr = pmc_traffic(args.dtype, B, N)
            out["phases_us_per_step"] = {k: 1e3 * v[0] / args.steps for k, v in phases.items()}
            out["kernels"] = kernel_table(phases, ab, tr[0] if tr else None, FUSED_NAMES)
            # the materialised backward sweep against HBM: SURVEY 8(d)'s contract figure (dense tensors in HBM)
            ab_m = mat["ab"]
            ms, n = mat["phases"]["backward"]
            avg_s = ms / max(n, 1) * 1e-3
            achieved = ab_m["backward"] / avg_s / 1e9
            roof_m = {"bound": "hbm", "kernel": "backward Riccati sweep over the materialised expansion (backward_tile16_kernel)",
                      "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                      "traffic": tr[0].get("backward") if tr else None, "traffic_source": tr[1] if tr else None,
                      "algorithmic_bytes_per_launch": ab_m["backward"], "avg_launch_us": avg_s * 1e6, "launches": n,
                      "avg_launch_us_back_to_back": mat["bwd_us_back_to_back"],
                      "frac_of_measured_copy_peak_6.29TBs": achieved / 6290.0}
            if args.materialised:
                out["roofline"] = roof_m
            else:
                # The fused kernel never materialises the expansion: its HBM bytes are the trajectory and the gains
                # ("a fused variant must report against its own smaller byte count and say so"), and what
                # bounds it is vector-instruction issue: every SIMD of the chip holds one sweep wave and two producer
                # waves, a wave64 instruction occupies its SIMD for 4 cycles.
                # (a shard of <= 1024 trajectories runs the whole iteration as ONE launch of ilqr_persistent_kernel:
                # the headline batch does not, a small --batch does)
                pk = "persist" if phases["persist"][1] and not phases["fused"][1] else "fused"
                kname = "ilqr_persistent_kernel" if pk == "persist" else "backward_fused16_kernel"
                ms_f, n_f = phases[pk]
                avg_f = ms_f / max(n_f, 1) * 1e-3
                vi = pmc_valu(args.dtype, B, N)
                insts = vi[0].get(kname) if vi else None
                peak_gips = 1024 * 2.4 / 4.0      # 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 cycles at 2.4 GHz
                ach = insts / avg_f / 1e9 if insts else None
                out["roofline"] = {
                    "bound": "valu", "kernel": ("ilqr_persistent_kernel (the whole iteration in one launch; no expansion in HBM)"
                                                if pk == "persist" else
                                                "backward_fused16_kernel (acceptance step + linearise + backward sweep; "
                                                "no expansion in HBM)"),
                    "achieved": ach, "peak": peak_gips, "unit": "G wave-instructions/s",
                    "frac": ach / peak_gips if ach else None,
                    "instructions_per_launch": insts, "instructions_source": vi[1] if vi else None,
                    "traffic": tr[0].get(pk) if tr else None, "traffic_source": tr[1] if tr else None,
                    "algorithmic_bytes_per_launch": ab[pk], "avg_launch_us": avg_f * 1e6, "launches": n_f,
                    "hbm_achieved_GBs_on_own_bytes": ab[pk] / avg_f / 1e9,
                    "hbm_frac_on_own_bytes": ab[pk] / avg_f / 1e9 / HBM_PEAK_GBS,
                    "note": "bound = vector-instruction issue (SQ_INSTS_VALU per launch / launch time against 1024 SIMDs x "
                            "2.4 GHz / 4 cycles); the HBM fraction on its own (small) byte count is given beside it; the "
                            "materialised sweep's HBM roofline is `roofline_materialised`"}
                out["roofline_materialised"] = roof_m
                out["materialised"] = {
                    "ms_per_step": mat["wall_s_per_step"] * 1e3, "value": B / mat["wall_s_per_step"],
                    "phases_us_per_step": {k: 1e3 * v[0] / args.steps for k, v in mat["phases"].items() if v[1]},
                    "kernels": kernel_table(mat["phases"], ab_m, tr[0] if tr else None, FUSED_NAMES),
                    "note": "the same K steps with ILQR_FLAG_NO_FUSE, outside the timed region"}
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
                                  device=local_rank, flags=path_flags, stream=stream)
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
            out["mpc_c4_shard"] = mpc_extra(ilqr_amd, _lib, problems, np_dt, local_rank, stream)
            out["configs"] = config_extras(ilqr_amd, _lib, problems, local_rank, stream)
        print(json.dumps(out))
    if world > 1 or args.exchange:
        rank_barrier(dist, local_rank)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
