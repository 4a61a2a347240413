"""Persistent kernel against the two-launch iteration at B = 4096: wall per iteration for calls of n iterations, same call
sequence on both (the cost of an iteration drifts with the iteration count in the fixed-iteration mode)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib, problems
p = problems.ua_double_pendulum()
B = int(os.environ.get("B", 4096))
x0, U0 = problems.ua_batch(B, seed=1000)
for tag, fl in (("persistent", 0), ("two launches", _lib.FLAG_NO_PERSIST)):
    h = ilqr_amd.make_system(p["dynamics"], p["cost"], np.float32).make_handle(horizon=200, batch=B, n_alpha=10, maxiter=1 << 30,
                                                                              flags=_lib.FLAG_KEEP_ITERATING | fl)
    h.set_problem(x0, U0); h.initial_rollout(); h.iterate(3); h.flush(); h.sync()
    out = []
    for n in (20, 20, 1, 1, 5, 100, 20):
        t0 = time.perf_counter(); h.iterate(n); h.flush(); h.sync(); w = time.perf_counter() - t0
        out.append(f"n={n}: {w / n * 1e6:.1f}")
    print(tag, " | ".join(out), flush=True)
    h.close()
