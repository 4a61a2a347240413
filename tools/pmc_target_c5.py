"""Small fixed workload for rocprofv3 --pmc passes: c5 shard (n=16, m=8, N=500, B=128 or ILQR_PMC_B), a few iterations."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib, problems
B = int(os.environ.get("ILQR_PMC_B", "128"))
dt = np.float64 if os.environ.get("ILQR_PMC_DTYPE", "f32") == "f64" else np.float32
p = problems.linear_quadratic()
x0, U0 = problems.lq_batch(B, 16, 8, 500)
h = ilqr_amd.make_system(p["dynamics"], p["cost"], dt).make_handle(horizon=500, batch=B, n_alpha=10, maxiter=1 << 30,
                                                                  flags=_lib.FLAG_KEEP_ITERATING)
h.set_problem(x0, U0); h.initial_rollout(); h.iterate(4); h.sync()
print("done")
