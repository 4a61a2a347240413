#!/usr/bin/env python3
"""Open-loop swing-up on the MI355X path: counterpart of the reference driver
python/run_iLQR_open_loop.py (parameters :16-43, system :52-59, solver :62-69, warm-up :78-93,
timed solve :104-108), running on ilqr_amd instead of class_files.  `--system ua` runs the
under-actuated double pendulum of python/run_iLQR_OL_UA_Pendulum.py (:16-79).

    python scripts/run_iLQR_open_loop.py [--system pendulum|ua] [--batch B] [--dtype f64|f32] [--plot out.png]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ilqr_amd.iLQR_class import iLQR                                  # noqa: E402
from ilqr_amd.systems.pendulum_sys import MyPendulum                   # noqa: E402
from ilqr_amd.systems.UA_double_pendulum_sys import MyUADoublePendulum  # noqa: E402


def build(kind, dtype):
    dt = 0.01
    if kind == "pendulum":
        T, tol, maxiter = 4.0, 1e-5, 100
        system = MyPendulum(dt=dt, x_target=np.array([np.pi, 0.0]), Q=np.diag([1.0, 1.0]), R=np.diag([1.0]),
                            Q_f=np.diag([0.0, 0.0]), g=9.81, l=1.0, d=0.0, integrator="backward_euler", dtype=dtype)
        x_0 = np.array([1.0, 0.0])
    else:
        T, tol, maxiter = 8.0, 1e-5, 700
        system = MyUADoublePendulum(dt=dt, x_target=np.array([np.pi, 0.0, 0.0, 0.0]), Q=np.diag([5.0, 5.0, 0.1, 0.1]),
                                    R=np.diag([50.0]), Q_f=np.diag([1000.0, 1000.0, 10.0, 10.0]), g=9.81, m1=1.0,
                                    m2=1.0, l1=1.0, l2=1.0, d1=0.1, d2=0.1, theta1=1.0 / 12.0, theta2=1.0 / 12.0,
                                    integrator="backward_euler", dtype=dtype)
        x_0 = np.zeros(4)
    N = len(np.arange(0, T + dt, dt)) - 1
    return system, T, N, x_0, np.zeros((system.n_u, N)), tol, maxiter


def main(argv=None):
    """Runs the driver; returns what it computed (the tests call this and compare with the oracle)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--system", default="pendulum", choices=["pendulum", "ua"])
    ap.add_argument("--batch", type=int, default=0, help="0 = the reference's single trajectory; B>0 = B perturbed starts")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--plot", default=None, help="write the reference's figure (run_iLQR_open_loop.py:115-145) to this file")
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args(argv)
    dtype = np.float64 if a.dtype == "f64" else np.float32
    print("Setting up parameters...")
    system, T, N, x_0, U_init, tol, maxiter = build(a.system, dtype)
    if a.batch:
        rng = np.random.default_rng(0)
        x_0 = x_0[None, :] + 0.1 * rng.standard_normal((a.batch, system.n_x))
        U_init = np.zeros((a.batch,) + U_init.shape)
    solver = iLQR(system=system, T=T, x_0=x_0, U_init=U_init, tol=tol, maxiter=maxiter, verbose=not a.quiet)

    print("Warming up ...")   # the reference warms up the JIT (:78-93); here it pages in the kernels
    Xw, Uw = np.zeros_like(solver.X), np.zeros_like(solver.U)
    solver.backward_pass(Xw, Uw)[0].block_until_ready()
    solver.forward_pass(solver.x_0, 0.0, Xw, Uw, np.zeros_like(solver.U_ff), np.zeros_like(solver.K))[0].block_until_ready()

    print("Running iLQR...")
    t0 = time.time()
    X_bar, U_bar, cost = solver.optimize_trajectory()
    dt_solve = time.time() - t0
    print(f"Time taken to execute iLQR: {dt_solve:.4f} seconds")
    if a.batch:
        print(f"final cost min/median/max: {np.min(cost):.4f} / {np.median(cost):.4f} / {np.max(cost):.4f}")
    else:
        print(f"final cost {cost:.6f}  status {solver.status}  iterations {solver.iterations}")
        print(f"final state {np.asarray(X_bar)[:, -1]}")
    if a.plot:
        print("Plotting results...")
        from _plots import open_loop_figure
        Xp = np.asarray(X_bar if not a.batch else X_bar[0])
        Up = np.asarray(U_bar if not a.batch else U_bar[0])
        open_loop_figure(a.plot, solver.tspan, Xp, Up)
        print("wrote", a.plot)
    print("\n--- Summary ---")
    print(f"iLQR solver time:      {dt_solve:.4f} seconds")
    return dict(X=np.asarray(X_bar), U=np.asarray(U_bar), cost=cost, status=solver.status, iterations=solver.iterations,
                K=np.asarray(solver.K), U_ff=np.asarray(solver.U_ff), N=N, seconds=dt_solve)


if __name__ == "__main__":
    main()
