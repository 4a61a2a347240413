"""Debug: the (4,2) system, persistent / fused / materialised forms compared field by field after various call patterns."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib, problems

FIELDS = (("X", _lib.X), ("U", _lib.U), ("K", _lib.K), ("U_ff", _lib.UFF), ("cost", _lib.COST), ("alpha", _lib.ALPHA),
          ("status", _lib.STATUS), ("iters", _lib.ITERS))
B, N = int(os.environ.get("B", 37)), 100
p = problems.double_pendulum(N=N)
rng = np.random.default_rng(4)
x0 = np.asarray(p["x0"])[None] + 0.1 * rng.standard_normal((B, 4))
U0 = 0.05 * rng.standard_normal((B, 2, N))
dt = np.float64 if os.environ.get("F64") else np.float32
sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], dt)

def mk():
    hs = [sysm.make_handle(horizon=N, batch=B, n_alpha=10, n_trials=10, flags=f, tol=p["tol"], maxiter=30)
          for f in (0, _lib.FLAG_NO_PERSIST, _lib.FLAG_NO_FUSE)]
    for h in hs:
        h.set_problem(x0, U0)
        h.initial_rollout()
    return hs

def cmp(hs, what):
    for name, f in FIELDS:
        v = [h.get(f) for h in hs]
        for k in (0, 1):
            if not np.array_equal(v[k], v[2], equal_nan=True):
                d = np.argwhere(v[k] != v[2])
                print(f"{what}: {name} form {k} != materialised at {len(d)} places, first {d[:3].tolist()} trajectories {sorted(set(d[:,0].tolist()))[:10]}")
    print(what, "compared", flush=True)

for pattern in ((1, 1, 1, 1), (2,), (1, 3), (4,)):
    hs = mk()
    for n in pattern:
        for h in hs:
            h.iterate(n)
        cmp(hs, f"pattern {pattern} after iterate({n})")

hs = mk()
for h in hs:
    h.iterate(4)
X = [h.get(_lib.X) for h in hs]
st = [h.get(_lib.STATUS) for h in hs]
it = [h.get(_lib.ITERS) for h in hs]
al = [h.get(_lib.ALPHA) for h in hs]
for b in (7, 8):
    print("traj", b, "status", [int(s[b]) for s in st], "iters", [int(s[b]) for s in it], "alpha", [float(s[b]) for s in al])
    for k in range(3):
        print("  form", k, X[k][b, :, :6].tolist())

if os.environ.get("F64"):
    hs = mk()
    for h in hs:
        h.iterate(1)
    X = [h.get(_lib.X) for h in hs]
    d = np.argwhere(X[1] != X[2])
    print("diff places", d.tolist()[:12])
    for b, i, t in d[:4]:
        print("traj", b, "comp", i, "t", t, [float(X[k][b, i, t]) for k in range(3)], "neighbours form1", X[1][b, i, t - 2:t + 3].tolist(), "form2", X[2][b, i, t - 2:t + 3].tolist())
    print("alpha", [h.get(_lib.ALPHA)[[0, 2, 8]].tolist() for h in hs])
