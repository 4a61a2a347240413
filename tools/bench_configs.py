"""Per-iteration time of the other BASELINE.json configs on one GPU (the bench.py line is c3).
c1 pendulum B=1; c2 UA B=256; c4 shard: UA MPC 1024 instances; c5 shard: LQ n=16 m=8 N=500, B=128."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib, problems


def time_iters(sysm, x0, U0, N, iters=10, n_alpha=10, dtype=np.float32):
    h = sysm.make_handle(horizon=N, batch=len(x0), n_alpha=n_alpha, maxiter=1 << 30, flags=_lib.FLAG_KEEP_ITERATING)
    h.set_problem(x0, U0); h.initial_rollout(); h.iterate(3); h.sync()
    h.timing_enable(True); h.timing_reset()
    t0 = time.perf_counter(); h.iterate(iters); h.sync(); wall = (time.perf_counter() - t0) / iters
    ph = {k: round(v[0] / iters * 1e3, 1) for k, v in h.timing_get().items()}
    return wall, ph


for dt in (np.float32, np.float64):
    name = np.dtype(dt).name
    p = problems.pendulum_open_loop(integrator="backward_euler", N=400)
    w, ph = time_iters(ilqr_amd.make_system(p["dynamics"], p["cost"], dt), p["x0"][None], p["U_init"][None], 400)
    print(f"c1 {name}: pendulum backward_euler N=400 B=1: {w*1e6:.0f} us/iteration {ph}")
    p = problems.ua_double_pendulum()
    for B, tag in ((256, "c2"), (1024, "c4-shard"), (4096, "c3")):
        x0, U0 = problems.ua_batch(B, seed=0)
        w, ph = time_iters(ilqr_amd.make_system(p["dynamics"], p["cost"], dt), x0, U0, 200)
        print(f"{tag} {name}: UA double pendulum rk4 N=200 B={B}: {w*1e6:.0f} us/iteration = {B/w/1e6:.2f} M it/s {ph}")
    p = problems.linear_quadratic()
    x0, U0 = problems.lq_batch(128, 16, 8, 500)
    w, ph = time_iters(ilqr_amd.make_system(p["dynamics"], p["cost"], dt), x0, U0, 500)
    print(f"c5-shard {name}: LQ n=16 m=8 N=500 B=128: {w*1e6:.0f} us/iteration = {128/w/1e3:.1f} k it/s {ph}")

# user-defined systems (generated plugins, systems/custom_sys.py) at the c3 shape
from ilqr_amd.systems.examples import bench_cases
for name, (sysm, N, B, x0c) in bench_cases().items():
    rng = np.random.default_rng(0)
    dt = sysm.dtype
    x0 = (x0c[None, :] + 0.05 * rng.standard_normal((B, sysm.n_x))).astype(dt)
    U0 = (0.1 * rng.standard_normal((B, sysm.n_u, N))).astype(dt)
    w, ph = time_iters(sysm, x0, U0, N)
    print(f"{name} N={N} B={B}: {w*1e6:.0f} us/iteration = {B/w/1e6:.2f} M it/s {ph}")
