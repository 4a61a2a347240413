"""Two handles (2048 trajectories each) iterating on their own streams, for a rocprofv3 --kernel-trace timeline."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib, problems
p = problems.ua_double_pendulum()
x0, U0 = problems.ua_batch(4096, seed=0)
sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], np.float32)
hs = []
for lo in (0, 2048):
    h = sysm.make_handle(horizon=200, batch=2048, n_alpha=10, maxiter=1 << 30, flags=_lib.FLAG_KEEP_ITERATING)
    h.set_problem(x0[lo:lo + 2048], U0[lo:lo + 2048]); h.initial_rollout(); h.iterate(2); h.sync()
    hs.append(h)
for h in hs: h.iterate(4)
for h in hs: h.sync()
print("done")
