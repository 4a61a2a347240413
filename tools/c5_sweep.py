"""c5 shard (n=16, m=8, N=500, B=128): per-kernel times of one iteration, back-to-back sweeps, and the fp32 error of the
sweep against the fp64 C oracle.   ILQR_BACKWARD_WAVE_LDS=1 selects the LDS form for an A/B."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib, problems
from oracle.c_oracle import COracle

p = problems.linear_quadratic()
for B in (128, 1024):
    x0, U0 = problems.lq_batch(B, 16, 8, 500)
    for dt in (np.float32, np.float64):
        h = ilqr_amd.make_system(p["dynamics"], p["cost"], dt).make_handle(horizon=500, batch=B, n_alpha=10, maxiter=1 << 30,
                                                                          flags=_lib.FLAG_KEEP_ITERATING)
        h.set_problem(x0, U0); h.initial_rollout(); h.iterate(2); h.linearize()
        for _ in range(3): h.backward()
        h.sync(); h.timing_enable(True); h.timing_reset()
        for _ in range(20): h.backward()
        ms, n = h.timing_get()["backward"]
        h.timing_reset(); h.iterate(5)
        ph = {k: round(v[0] / 5 * 1e3, 1) for k, v in h.timing_get().items()}
        bytes_alg = h.algorithmic_bytes()["backward"]
        print(f"B={B} {np.dtype(dt).name}: sweep {ms / n * 1e3:7.1f} us back-to-back = {ms / n / 500 * 1e6:6.0f} ns/step, "
              f"{bytes_alg / (ms / n * 1e-3) / 1e9:6.0f} GB/s algorithmic; iteration {ph}", flush=True)
        h.close()
