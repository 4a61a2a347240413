import os, sys
import numpy as np
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
import ilqr_amd
from ilqr_amd import _lib, problems
p = problems.ua_double_pendulum(N=200)
B = int(os.environ.get("DBG_B", "1040"))
x0, U0 = problems.ua_batch(B, seed=5, restarts=True, N=200)
sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], np.float32)
hs = [sysm.make_handle(horizon=200, batch=B, n_alpha=10, n_trials=10, tol=p["tol"], maxiter=50, flags=f) for f in (int(os.environ.get("DBG_F0", "0")), int(os.environ.get("DBG_F1", "4")))]
for h in hs:
    h.set_problem(x0, U0); h.initial_rollout(); h.iterate(1)
for name, f in (("K", _lib.K), ("Uff", _lib.UFF), ("trial", _lib.TRIAL_COSTS), ("cost", _lib.COST), ("alpha", _lib.ALPHA), ("status", _lib.STATUS), ("X", _lib.X), ("U", _lib.U)):
    a, b = hs[0].get(f), hs[1].get(f)
    bad = ~np.isclose(a, b, rtol=0, atol=0, equal_nan=True)
    rows = np.unique(np.nonzero(bad)[0])
    print(name, "equal" if not bad.any() else f"{bad.sum()} of {bad.size} differ; trajectories {rows[:12]}... n={len(rows)}")
    if name == "trial" and bad.any():
        r = rows[0]; print("  persist", a[r]); print("  ref    ", b[r])
