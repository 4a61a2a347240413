"""Per-iteration time of the systems that do not run on the n=4/m=1 DPP sweep: pendulum (2,1), fully actuated
double pendulum (4,2), at batch 4096."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib, problems

def run(name, p, B, N, dt, x0, U0):
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], dt)
    h = sysm.make_handle(horizon=N, batch=B, n_alpha=10, maxiter=1 << 30, flags=_lib.FLAG_KEEP_ITERATING)
    h.set_problem(x0.astype(dt), U0.astype(dt)); h.initial_rollout(); h.iterate(3); h.sync()
    t0 = time.perf_counter(); h.iterate(10); h.sync(); wall = (time.perf_counter() - t0) / 10
    h.timing_enable(True); h.timing_reset(); h.iterate(10); h.sync()
    ph = {k: round(v[0] / 10 * 1e3, 1) for k, v in h.timing_get().items()}
    print(f"{name} {np.dtype(dt).name} B={B} N={N}: {wall*1e6:.0f} us/iter = {B/wall/1e6:.2f} M it/s {ph}", flush=True)
    h.close()

rng = np.random.default_rng(0)
B = 4096
for dt in (np.float32, np.float64):
    p = problems.pendulum_open_loop(N=200, integrator="rk4")
    run("pendulum rk4", p, B, 200, dt, np.asarray(p["x0"])[None] + 0.1 * rng.standard_normal((B, 2)), np.zeros((B, 1, 200)))
    p = problems.double_pendulum(N=100)
    run("double pendulum (4,2) rk4", p, B, 100, dt, np.asarray(p["x0"])[None] + 0.1 * rng.standard_normal((B, 4)), np.zeros((B, 2, 100)))
