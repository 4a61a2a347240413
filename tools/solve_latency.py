"""Solve-to-convergence wall time (ilqr_solve: device loop + the host's active-count read-back) and a short MPC run."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib, problems

p = problems.ua_double_pendulum()
for B in (1, 256, 4096):
    x0, U0 = problems.ua_batch(B, seed=0)
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], np.float32)
    s = ilqr_amd.iLQR(sysm, None, x0.astype(np.float32), U0.astype(np.float32), N=200, tol=1e-5, maxiter=50, verbose=False)
    s.optimize_trajectory()
    best = 1e9
    for rep in range(3):
        s._h.set_problem(x0.astype(np.float32), U0.astype(np.float32))
        t0 = time.perf_counter(); s.optimize_trajectory(); best = min(best, time.perf_counter() - t0)
    it = np.asarray(s.iterations)
    print(f"solve B={B}: {best*1e3:.2f} ms, iterations max {it.max()} mean {it.mean():.1f} -> {best/it.max()*1e6:.0f} us per loop iteration", flush=True)
pm = problems.ua_double_pendulum()
B = 1024
x0, U0 = problems.ua_batch(B, seed=2)
st = ilqr_amd.mpc_init(pm["dynamics"], pm["cost"], x0.astype(np.float32), U0.astype(np.float32), plant_integrator="backward_euler", N=200, tol=1e-5, maxiter=50, dtype=np.float32)
st.solver.mpc_run(2)
t0 = time.perf_counter(); st.solver.mpc_run(10); el = time.perf_counter() - t0
print(f"MPC B={B}: {el/10*1e3:.2f} ms per receding-horizon step", flush=True)
