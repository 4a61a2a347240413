"""Small fixed workload for rocprofv3 --pmc passes: c3 shape, a few iterations (env ILQR_PMC_B / ILQR_PMC_DTYPE)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib, problems
B = int(os.environ.get("ILQR_PMC_B", "4096"))
dt = np.float64 if os.environ.get("ILQR_PMC_DTYPE", "f32") == "f64" else np.float32
p = problems.ua_double_pendulum()
x0, U0 = problems.ua_batch(B, seed=0)
h = ilqr_amd.make_system(p["dynamics"], p["cost"], dt).make_handle(horizon=200, batch=B, n_alpha=10, maxiter=1 << 30,
                                                                  flags=_lib.FLAG_KEEP_ITERATING)
h.set_problem(x0, U0); h.initial_rollout(); h.iterate(4); h.sync()
print("done")
