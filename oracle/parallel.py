"""Whole batches through the C oracle on several host cores -- TEST INFRASTRUCTURE (see oracle/__init__.py).

One trajectory takes the C restatement a few milliseconds; a BASELINE-size batch (4096) takes ~30 s on one core and a
couple of seconds on the GPU box's 16.  The workers are separate `python oracle/parallel.py ...` processes started
with subprocess (not fork: the calling test process has usually initialised the GPU, and a forked copy of a process
with a live HIP runtime is not safe to run; not multiprocessing's spawn either: it re-imports the caller's main
module).  They exchange .npz files, import only NumPy and the C oracle, and never touch the GPU."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _solve_range(dynamics, cost, dtype_name, x0, U0, tol, maxiter):
    from oracle.c_oracle import COracle
    co = COracle(dynamics, cost, dtype=np.dtype(dtype_name))
    out = []
    for b in range(len(x0)):
        r = co.solve(x0[b], U0[b], tol=tol, maxiter=maxiter)
        out.append(dict(status=r["status"], iterations=r["iterations"], cost=float(r["cost"]),
                        alphas=r["alphas"], costs=np.asarray(r["costs"], np.float64)))
    return out


def _pack(results, maxiter):
    n = len(results)
    alphas, costs = np.zeros((n, max(maxiter, 1))), np.zeros((n, max(maxiter, 1)))
    for i, r in enumerate(results):
        alphas[i, : r["iterations"]] = r["alphas"]
        costs[i, : r["iterations"]] = r["costs"]
    return dict(status=np.array([r["status"] for r in results]), iterations=np.array([r["iterations"] for r in results]),
                cost=np.array([r["cost"] for r in results]), alphas=alphas, costs=costs)


def _unpack(z):
    out = []
    for i in range(len(z["cost"])):
        it = int(z["iterations"][i])
        out.append(dict(status=str(z["status"][i]), iterations=it, cost=float(z["cost"][i]),
                        alphas=z["alphas"][i, :it].copy(), costs=z["costs"][i, :it].copy()))
    return out


def _to_jsonable(d):
    return {k: (np.asarray(v).tolist() if isinstance(v, (np.ndarray, list, tuple)) else v) for k, v in d.items()}


def solve_many(dynamics, cost, x0, U0, dtype=np.float64, tol=1e-5, maxiter=100, procs=None):
    """optimize_trajectory (iLQR_class.py:250-313) of every (x0[b], U0[b]); returns a list of dicts with status,
    iterations, cost, alphas (accepted alpha per iteration, 0 = none) and costs (after each iteration)."""
    from oracle.c_oracle import build
    build()                                   # before the workers start: they only load the library
    if procs is None:
        try:
            procs = len(os.sched_getaffinity(0))
        except AttributeError:
            procs = os.cpu_count() or 1
        procs = max(1, min(procs, 16, len(x0) // 32 or 1))
    name = np.dtype(dtype).name
    if procs == 1:
        return _solve_range(dynamics, cost, name, x0, U0, tol, maxiter)
    chunks = [c for c in np.array_split(np.arange(len(x0)), procs) if len(c)]
    with tempfile.TemporaryDirectory(prefix="ilqr_oracle_") as d:
        spec = dict(dynamics=_to_jsonable(dynamics), cost=_to_jsonable(cost), dtype=name, tol=tol, maxiter=maxiter)
        json.dump(spec, open(os.path.join(d, "spec.json"), "w"))
        jobs = []
        for k, c in enumerate(chunks):
            np.savez(os.path.join(d, f"in{k}.npz"), x0=x0[c], U0=U0[c])
            jobs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), d, str(k)], cwd=ROOT))
        rcs = [j.wait() for j in jobs]
        if any(rcs):
            raise RuntimeError(f"C-oracle worker failed (exit codes {rcs})")
        out = []
        for k in range(len(chunks)):
            out += _unpack(np.load(os.path.join(d, f"out{k}.npz")))
    return out


if __name__ == "__main__":          # worker: parallel.py <dir> <k>
    sys.path.insert(0, ROOT)
    d, k = sys.argv[1], sys.argv[2]
    spec = json.load(open(os.path.join(d, "spec.json")))
    dyn = {key: (np.array(v) if isinstance(v, list) else v) for key, v in spec["dynamics"].items()}
    cst = {key: np.array(v) for key, v in spec["cost"].items()}
    z = np.load(os.path.join(d, f"in{k}.npz"))
    res = _solve_range(dyn, cst, spec["dtype"], z["x0"], z["U0"], spec["tol"], spec["maxiter"])
    np.savez(os.path.join(d, f"out{k}.npz"), **_pack(res, spec["maxiter"]))
