"""Small fixed workload for rocprofv3 passes: BASELINE config c4 at one GPU's shard -- 1024 warm-started MPC instances of
the under-actuated double pendulum (N = 200, rk4 optimiser, backward_euler plant, tol 1e-5, maxiter 50), 2 cold + 6 warm
receding-horizon steps device-resident (ilqr_mpc_run).  ILQR_PMC_DTYPE selects f32 / f64."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib, problems
B = int(os.environ.get("ILQR_PMC_B", "1024"))
dt = np.float64 if os.environ.get("ILQR_PMC_DTYPE", "f32") == "f64" else np.float32
p = problems.ua_double_pendulum(N=200)
x0, U0 = problems.ua_batch(B, seed=2, restarts=False, N=200)
h = ilqr_amd.make_system(p["dynamics"], p["cost"], dt).make_handle(horizon=200, batch=B, n_alpha=10, n_trials=10, tol=p["tol"],
                                                                  maxiter=p["maxiter"], plant_integrator="backward_euler")
h.mpc_reset(x0, U0)
h.mpc_run(2)
u, x, c = h.mpc_run(6)
print("done", bool(np.isfinite(c).all()), float(np.mean(h.get(_lib.ITERS))))
