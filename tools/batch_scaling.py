"""Per-iteration time and phase breakdown vs batch size (single handle): shows where the sequential kernels stop
being lone-wave latency bound."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib, problems

p = problems.ua_double_pendulum()
N, iters = 200, 10
for dt in (np.float32,):
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], dt)
    for B in (1024, 2048, 4096, 8192, 16384, 32768):
        x0, U0 = problems.ua_batch(B, seed=0)
        h = sysm.make_handle(horizon=N, batch=B, n_alpha=10, maxiter=1 << 30, flags=_lib.FLAG_KEEP_ITERATING)
        h.set_problem(x0, U0); h.initial_rollout(); h.iterate(3); h.sync()
        t0 = time.perf_counter(); h.iterate(iters); h.sync(); wall = (time.perf_counter() - t0) / iters
        h.timing_enable(True); h.timing_reset(); h.iterate(iters); h.sync()
        ph = {k: round(v[0] / iters * 1e3, 1) for k, v in h.timing_get().items()}
        print(f"{np.dtype(dt).name} B={B}: {wall*1e6:.0f} us/iter = {B/wall/1e6:.2f} M it/s {ph}", flush=True)
        h.close()
