"""Experiment: does splitting the 4096-trajectory batch over S concurrent streams (S handles of 4096/S trajectories)
raise whole-batch throughput?  The sequential kernels are lone-wave latency bound at one wave per SIMD, so
independent sub-batches in different phases should interleave on the same SIMDs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib, problems

p = problems.ua_double_pendulum()
B, N, iters = 4096, 200, 20
for dt in (np.float32, np.float64):
    x0, U0 = problems.ua_batch(B, seed=0)
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], dt)
    for S in (1, 2, 4):
        edges = np.linspace(0, B, S + 1).astype(int)
        hs = []
        for s in range(S):
            lo, hi = edges[s], edges[s + 1]
            h = sysm.make_handle(horizon=N, batch=hi - lo, n_alpha=10, maxiter=1 << 30, flags=_lib.FLAG_KEEP_ITERATING)
            h.set_problem(x0[lo:hi], U0[lo:hi]); h.initial_rollout(); h.iterate(3)
            hs.append(h)
        for h in hs: h.sync()
        best = best2 = alone = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for k in range(iters):          # interleave the launches so no stream runs ahead of the others on the host
                for h in hs: h.iterate(1)
            for h in hs: h.sync()
            best = min(best, (time.perf_counter() - t0) / iters)
            t0 = time.perf_counter()
            for h in hs: h.iterate(iters)   # deep queues, one call per stream
            t1 = time.perf_counter()
            for h in hs: h.sync()
            best2 = min(best2, (time.perf_counter() - t0) / iters)
            t0 = time.perf_counter()
            hs[0].iterate(iters); hs[0].sync()
            alone = min(alone, (time.perf_counter() - t0) / iters)
        print(f"{np.dtype(dt).name} S={S}: interleaved {best*1e6:.0f} us, deep queues {best2*1e6:.0f} us (enqueue {1e6*(t1-t0+0)/iters:.0f}), "
              f"one sub-batch alone {alone*1e6:.0f} us", flush=True)
        for h in hs: h.close()
