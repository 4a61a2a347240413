"""Back-to-back backward sweeps vs batch size (fp32, N = 200): per-step time of one wave when the tile tensor fits the
L2 (B <= 512: 20 MB), the Infinity Cache (B <= 4096: 157 MB) or neither.  Separates memory latency from the step's own
instruction stream."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib, problems

p = problems.ua_double_pendulum()
N, R = 200, 40
DTS = [np.dtype(a).type for a in sys.argv[1].split(',')] if len(sys.argv) > 1 else (np.float32, np.float64)
BS = [int(a) for a in sys.argv[2].split(',')] if len(sys.argv) > 2 else (64, 256, 512, 1024, 2048, 4096, 8192)
for dt in DTS:
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], dt)
    for B in BS:
        x0, U0 = problems.ua_batch(B, seed=0)
        h = sysm.make_handle(horizon=N, batch=B, n_alpha=10, maxiter=1 << 30, flags=_lib.FLAG_KEEP_ITERATING)
        h.set_problem(x0, U0); h.initial_rollout(); h.iterate(2); h.linearize()
        for _ in range(3): h.backward()
        h.sync()
        h.timing_enable(True); h.timing_reset()
        for _ in range(R): h.backward()
        ms, n = h.timing_get()["backward"]
        us = ms / n * 1e3
        tiles_mb = N * B * 48 * np.dtype(dt).itemsize / 1e6
        print(f"{np.dtype(dt).name} B={B:5d} tiles {tiles_mb:7.1f} MB: {us:7.2f} us/launch = {us / N * 1e3:6.1f} ns/step", flush=True)
        h.close()
