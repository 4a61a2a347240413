#!/usr/bin/env python3
"""Receding-horizon MPC on the MI355X path: counterpart of the reference drivers
python/run_iLQR_MPC.py (pendulum; horizon :20-22, costs :39-41, optimiser model backward_euler :58-65,
plant midpoint :68-75, warm-up = one full solve :95, loop :116-143) and python/run_iLQR_UA_MPC.py
(under-actuated double pendulum: rk4 optimiser, backward_euler plant, maxiter 50, :17-174).

Two modes: `--host-loop` replays the reference loop statement by statement on the mirrored class
surface (x_0 / U attribute writes, optimize_trajectory, plant f_fcn, shift); the default keeps the whole
loop on the device (ilqr_mpc_run) and can carry a batch of independent MPC instances.

    python scripts/run_iLQR_MPC.py [--system pendulum|ua] [--batch B] [--steps K] [--host-loop] [--plot out.png]
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
    dt, T_h = 0.01, 2.0
    if kind == "pendulum":
        kw = dict(dt=dt, x_target=np.array([np.pi, 0.0]), Q=np.diag([10.0, 1.0]), R=np.diag([1.0]),
                  Q_f=np.diag([10.0, 10.0]), g=9.81, l=1.0, d=0.0, dtype=dtype)
        return (MyPendulum(integrator="backward_euler", **kw), MyPendulum(integrator="midpoint", **kw), T_h, 4.0,
                np.zeros(2), 1e-5, 10)
    kw = dict(dt=dt, x_target=np.array([np.pi, 0.0, 0.0, 0.0]), Q=np.diag([5.0, 5.0, 0.1, 0.1]), R=np.diag([50.0]),
              Q_f=np.diag([1000.0, 1000.0, 10.0, 10.0]), g=9.81, m1=1.0, m2=1.0, l1=1.0, l2=1.0, d1=0.1, d2=0.1,
              theta1=1.0 / 12.0, theta2=1.0 / 12.0, dtype=dtype)
    return (MyUADoublePendulum(integrator="rk4", **kw), MyUADoublePendulum(integrator="backward_euler", **kw), T_h, 5.0,
            np.zeros(4), 1e-5, 50)


def main(argv=None):
    """Runs the driver; returns what it computed (the tests call this and compare with the oracle)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--system", default="pendulum", choices=["pendulum", "ua"])
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--steps", type=int, default=0, help="MPC steps (default: the reference's T_sim / dt)")
    ap.add_argument("--horizon", type=float, default=0.0, help="horizon in seconds (default: the reference's 2.0)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--host-loop", action="store_true")
    ap.add_argument("--plot", default=None, help="write the reference's figure (run_iLQR_MPC.py:150-186) to this file")
    a = ap.parse_args(argv)
    dtype = np.float64 if a.dtype == "f64" else np.float32
    print("Setting up MPC parameters...")
    sys_opt, sys_sim, T_h, T_sim, x_0, tol, maxiter = build(a.system, dtype)
    if a.horizon:
        T_h = a.horizon
    dt = sys_opt.dt
    N_h = len(np.arange(0, T_h + dt, dt)) - 1
    N_sim = a.steps or (len(np.arange(0, T_sim + dt, dt)) - 1)
    U_init = np.zeros((sys_opt.n_u, N_h))
    if a.batch:
        rng = np.random.default_rng(2)
        x_0 = x_0[None, :] + rng.standard_normal((a.batch, sys_opt.n_x)) * 0.05
        U_init = np.zeros((a.batch,) + U_init.shape)
    solver = iLQR(system=sys_opt, T=T_h, x_0=x_0, U_init=U_init, tol=tol, maxiter=maxiter, verbose=False,
                  plant=sys_sim)

    print("Warming up ...")
    if a.system == "pendulum":
        # run_iLQR_MPC.py:95 -- the warm-up is ONE FULL SOLVE on the solver object.  It leaves X, K, U_ff behind, and
        # the loop's first optimize_trajectory() starts from them (alpha = 0 rollout through the warm-up's gains,
        # iLQR_class.py:257-259): part of the reference's closed-loop result, not just a timing detail.
        solver.optimize_trajectory()[0].block_until_ready()
    else:
        # run_iLQR_UA_MPC.py:114-124 -- the pure functions only; the solver state stays as constructed
        Xw, Uw = np.zeros_like(solver.X), np.zeros_like(solver.U)
        solver.backward_pass(Xw, Uw)[0].block_until_ready()
        solver.forward_pass(solver.x_0, 0.0, Xw, Uw, np.zeros_like(solver.U_ff), np.zeros_like(solver.K))[0].block_until_ready()
    print("Warm-up complete.")

    print("Running MPC simulation...")
    t0 = time.time()
    if a.host_loop:
        if a.batch:
            raise SystemExit("--host-loop replays the reference's single-instance loop; drop --batch")
        X_sim = np.zeros((sys_opt.n_x, N_sim + 1))
        U_sim = np.zeros((sys_opt.n_u, N_sim))
        costs = np.zeros(N_sim)
        current_x, U_guess = x_0, U_init
        X_sim[:, 0] = current_x
        for k in range(N_sim):
            solver.x_0 = current_x                                   # run_iLQR_MPC.py:118
            solver.U = U_guess                                       # :121
            X_bar, U_bar, cost = solver.optimize_trajectory()        # :124
            uk = U_bar[:, 0]                                         # :127
            x_next = sys_sim.f_fcn(current_x, uk)                    # :130
            U_sim[:, k], X_sim[:, k + 1], costs[k] = uk, x_next, cost
            U_guess = np.concatenate([U_bar[:, 1:], U_bar[:, -1:]], axis=1)   # :137
            current_x = x_next
            if k % 100 == 0:
                print(f"MPC Step {k}/{N_sim}...")
        x_end = X_sim[:, -1]
    else:
        # the same loop on the device; keep_state carries the warm-up solve's X, K, U_ff into step 0 (pendulum driver)
        solver.mpc_reset(x_0, U_init, keep_state=(a.system == "pendulum"))
        U_dev, X_dev, costs = solver.mpc_run(N_sim)       # (N_sim, [B,] n_u), (N_sim, [B,] n_x): state AFTER each step
        x_end = X_dev[-1]
        if not a.batch:                                   # the reference's layouts: X_sim (n_x, N_sim+1), U_sim (n_u, N_sim)
            X_sim = np.concatenate([np.asarray(x_0)[:, None], np.asarray(X_dev).T], axis=1)
            U_sim = np.asarray(U_dev).T
        else:
            X_sim, U_sim = X_dev, U_dev
    el = time.time() - t0
    print("MPC simulation finished.")
    print(f"Total MPC time: {el:.4f} seconds")
    print(f"Average time per step: {el / N_sim:.5f} seconds")
    print("final plant state:", np.asarray(x_end) if not a.batch else np.asarray(x_end)[:3])
    if a.plot:
        print("Plotting results...")
        from _plots import closed_loop_figure
        if a.batch:     # instance 0 of the batch
            Xp = np.concatenate([np.asarray(x_0)[0][:, None], np.asarray(X_sim)[:, 0].T], axis=1)
            Up = np.asarray(U_sim)[:, 0].T
        else:
            Xp, Up = X_sim, U_sim
        closed_loop_figure(a.plot, np.arange(N_sim + 1) * dt, Xp, Up, sys_opt.x_target)
        print("wrote", a.plot)
    return dict(X_sim=np.asarray(X_sim), U_sim=np.asarray(U_sim), cost=np.asarray(costs), N_h=N_h, N_sim=N_sim, seconds=el)


if __name__ == "__main__":
    main()
