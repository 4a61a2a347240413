"""Diagnostic: the device's fp32 solve (the bench dtype) at the c3 shape against the C oracle in fp32 AND fp64,
trajectory by trajectory -- final status, iteration count, accepted-alpha sequence, final cost.

With cost ~ 3e3 and tol = 1e-5 the reference's stopping rules (|dcost| <= tol, iLQR_class.py:267; cost_new <= cost,
:289) fall below one fp32 ulp of the cost (2.4e-4), so in the reference's own precision the LAST iterations of a
solve are decided by rounding, and two fp32 implementations with different operation orders stop at different
iterations.  This tool measures how far the agreement goes; tests/test_gpu_fullshape.py asserts the bounds.

    python tools/f32_status_parity.py [--batch 4096] [--horizon 200] [--sample 4096] [--out file.json]
"""
import argparse
import json
import multiprocessing as mp
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

STAT = {0: "active", 1: "converged", 2: "linesearch_failed", 3: "maxiter"}


def oracle_worker(args):
    p, dtype_name, x0, U0, tol, maxiter = args
    from oracle.c_oracle import COracle
    co = COracle(p["dynamics"], p["cost"], dtype=np.dtype(dtype_name))
    out = []
    for b in range(len(x0)):
        r = co.solve(x0[b], U0[b], tol=tol, maxiter=maxiter)
        out.append((r["status"], r["iterations"], float(r["cost"]), r["alphas"].tolist(), np.asarray(r["costs"], np.float64).tolist()))
    return out


def run_oracle(p, dtype_name, x0, U0, tol, maxiter, procs):
    chunks = np.array_split(np.arange(len(x0)), procs)
    with mp.get_context("fork").Pool(procs) as pool:
        res = pool.map(oracle_worker, [(p, dtype_name, x0[c], U0[c], tol, maxiter) for c in chunks if len(c)])
    return [r for part in res for r in part]


def device_trace(p, x0, U0, tol, maxiter, dtype):
    """Per-iteration (alpha, cost, status) of the device solve: the stage API stepped one iteration at a time
    (identical kernels and decisions to ilqr_solve; only the host reads in between)."""
    import ilqr_amd
    from ilqr_amd import _lib
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], dtype)
    h = sysm.make_handle(horizon=U0.shape[2], batch=len(x0), n_alpha=10, n_trials=10, tol=tol, maxiter=maxiter)
    h.set_problem(x0, U0)
    h.initial_rollout()
    c0 = h.get(_lib.COST).astype(np.float64)
    alphas, costs = [], []
    for _ in range(maxiter):
        st = h.get(_lib.STATUS) & 0xff
        if not (st == 0).any():
            break
        h.iterate(1)
        alphas.append(h.get(_lib.ALPHA).astype(np.float64))
        costs.append(h.get(_lib.COST).astype(np.float64))
    st = h.get(_lib.STATUS) & 0xff
    its = h.get(_lib.ITERS)
    cost = h.get(_lib.COST).astype(np.float64)
    return dict(status=st, iters=its, cost=cost, alphas=np.array(alphas), costs=np.array(costs), cost0=c0)


def compare(dev, orc, label):
    B = len(orc)
    st_o = np.array([{"converged": 1, "linesearch_failed": 2, "maxiter": 3}[r[0]] for r in orc])
    it_o = np.array([r[1] for r in orc])
    c_o = np.array([r[2] for r in orc])
    st_d, it_d, c_d = dev["status"][:B], dev["iters"][:B], dev["cost"][:B]
    same = (st_o == st_d) & (it_o == it_d)
    prefix, first_div_margin = [], []
    for b in range(B):
        a_o = np.array(orc[b][3])
        a_d = dev["alphas"][: it_d[b], b]
        n = min(len(a_o), len(a_d))
        k = 0
        while k < n and a_o[k] == a_d[k]:
            k += 1
        prefix.append(k)
        if k < max(len(a_o), len(a_d)) and k < len(orc[b][4]):
            # relative cost change of the oracle at the first iteration where the two traces part
            prev = orc[b][4][k - 1] if k > 0 else float(dev["cost0"][b])
            first_div_margin.append(abs(prev - orc[b][4][k]) / abs(prev))
    rel = np.abs(c_d - c_o) / np.abs(c_o)
    out = {
        "n": int(B),
        "status_counts_device": {STAT[k]: int((st_d == k).sum()) for k in (1, 2, 3)},
        "status_counts_oracle": {STAT[k]: int((st_o == k).sum()) for k in (1, 2, 3)},
        "iters_mean_device": float(it_d.mean()), "iters_mean_oracle": float(it_o.mean()),
        "same_status_and_iters_frac": float(same.mean()),
        "same_status_frac": float((st_o == st_d).mean()),
        "iters_absdiff_percentiles_50_90_99_max": [float(np.percentile(np.abs(it_d - it_o), q)) for q in (50, 90, 99, 100)],
        "alpha_prefix_equal_full_frac": float(np.mean([prefix[b] == min(it_d[b], it_o[b]) for b in range(B)])),
        "alpha_prefix_len_mean": float(np.mean(prefix)),
        "first_divergence_oracle_rel_dcost_percentiles_50_90_99_max":
            [float(np.percentile(first_div_margin, q)) for q in (50, 90, 99, 100)] if first_div_margin else None,
        "final_cost_rel_diff_percentiles_50_90_99_max": [float(np.percentile(rel, q)) for q in (50, 90, 99, 100)],
    }
    print(label, json.dumps(out, indent=1))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=200)
    ap.add_argument("--sample", type=int, default=4096)
    ap.add_argument("--maxiter", type=int, default=50)
    ap.add_argument("--procs", type=int, default=16)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from ilqr_amd import problems
    p = problems.ua_double_pendulum(N=a.horizon)
    x0, U0 = problems.ua_batch(a.batch, seed=1000, N=a.horizon)
    S = min(a.sample, a.batch)
    # oracles first: their workers are forked before this process touches the GPU
    x0r = x0.astype(np.float32).astype(np.float64)   # both sides see the fp32-rounded inputs
    o32 = run_oracle(p, "float32", x0r[:S], U0[:S], p["tol"], a.maxiter, a.procs)
    o64 = run_oracle(p, "float64", x0r[:S], U0[:S], p["tol"], a.maxiter, a.procs)
    res = {}
    d32 = device_trace(p, x0, U0, p["tol"], a.maxiter, np.float32)
    res["device_f32_vs_oracle_f32"] = compare(d32, o32, "device f32 vs C oracle f32")
    res["device_f32_vs_oracle_f64"] = compare(d32, o64, "device f32 vs C oracle f64")
    # how far do two fp32/fp64 runs of the SAME restatement agree?  (the floor any fp32 implementation faces)
    fake = dict(status=np.array([{"converged": 1, "linesearch_failed": 2, "maxiter": 3}[r[0]] for r in o32]),
                iters=np.array([r[1] for r in o32]), cost=np.array([r[2] for r in o32]),
                alphas=np.array([[r[3][i] if i < len(r[3]) else 0.0 for r in o32] for i in range(a.maxiter)]),
                cost0=d32["cost0"][:S])
    res["oracle_f32_vs_oracle_f64"] = compare(fake, o64, "C oracle f32 vs C oracle f64")
    d64 = device_trace(p, x0r, U0, p["tol"], a.maxiter, np.float64)
    res["device_f64_vs_oracle_f64"] = compare(d64, o64, "device f64 vs C oracle f64")
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
