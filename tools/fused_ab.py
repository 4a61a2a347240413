"""Fused iteration (backward_fused16_kernel + rollout) against the materialised one (linearise, sweep, rollout, select)
at the UA double pendulum shapes: wall per iteration and per-kernel HIP-event times."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib, problems

p = problems.ua_double_pendulum()
dts = [np.float32, np.float64] if "--f64" in sys.argv else [np.float32]
Bs = [int(v) for v in os.environ.get("FUSED_AB_B", "4096,1024,256").split(",")]
paths = ((("persistent", 0), ("fused", _lib.FLAG_NO_PERSIST)) if "--fused-only" in sys.argv else
         (("persistent", 0), ("fused", _lib.FLAG_NO_PERSIST), ("materialised", _lib.FLAG_NO_FUSE)))
for dt in dts:
    for B in Bs:
        x0, U0 = problems.ua_batch(B, seed=0)
        for tag, fl in paths:
            h = ilqr_amd.make_system(p["dynamics"], p["cost"], dt).make_handle(
                horizon=200, batch=B, n_alpha=10, maxiter=1 << 30, flags=_lib.FLAG_KEEP_ITERATING | fl)
            h.set_problem(x0, U0); h.initial_rollout(); h.iterate(5); h.sync()
            t0 = time.perf_counter(); h.iterate(20); h.sync(); wall = (time.perf_counter() - t0) / 20      # (persistent: ONE launch of 20 iterations)
            t0 = time.perf_counter()
            for _ in range(20): h.iterate(1)
            h.sync(); wall1 = (time.perf_counter() - t0) / 20
            h.timing_enable(True); h.timing_reset(); h.iterate(20); h.flush()
            ph = {k: round(v[0] / 20 * 1e3, 1) for k, v in h.timing_get().items() if v[1]}
            print(f"{np.dtype(dt).name} B={B} {tag}: {wall*1e6:.1f} us/iteration = {B/wall/1e6:.2f} M it/s (one call per iteration: {wall1*1e6:.1f} us) {ph}", flush=True)
            h.close()
