"""Solve-to-convergence and MPC step time, persistent kernel against the host-looped launches, at several batch sizes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib, problems
p = problems.ua_double_pendulum(N=200)
for B in [int(v) for v in os.environ.get("AB_B", "1024,4096,8192").split(",")]:
    x0, U0 = problems.ua_batch(B, seed=2, N=200)
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], np.float32)
    for tag, fl in (("persistent", 0), ("launches", _lib.FLAG_NO_PERSIST)):
        h = sysm.make_handle(horizon=200, batch=B, n_alpha=10, n_trials=10, tol=p["tol"], maxiter=50, plant_integrator="backward_euler", flags=fl)
        h.set_problem(x0, U0); h.solve(); h.set_problem(x0, U0); h.sync()
        t0 = time.perf_counter(); its, _ = h.solve(); ts = time.perf_counter() - t0
        h.mpc_reset(x0, U0); h.mpc_run(2)
        t0 = time.perf_counter(); h.mpc_run(10); tm = (time.perf_counter() - t0) / 10
        print(f"B={B} {tag}: solve {ts*1e3:.2f} ms (mean {its.mean():.1f} it), MPC {tm*1e3:.3f} ms/step = {B/tm/1e3:.0f} k instance-steps/s", flush=True)
        h.close()
