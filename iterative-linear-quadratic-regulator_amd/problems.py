"""Synthetic problem definitions used by tests and bench.py (BASELINE.json configs).

Pure data: parameter dictionaries (taken from the reference drivers) and seeded input
generators.  No numerics live here.

  c1  pendulum, n=2 m=1, N=100 (BASELINE) / N=400 (run_iLQR_open_loop.py:16-43), batch 1
  c2  under-actuated double pendulum, n=4 m=1, N=200, batch 256 (run_iLQR_UA_MPC.py:17-67)
  c3  same, batch 4096, 8 parallel line-search alphas
  c4  same, 8192 warm-started MPC instances over 8 GPUs
  c5  synthetic linear-quadratic system n=16 m=8 N=500, batch 1024 over 8 GPUs
"""

from __future__ import annotations

import numpy as np


def pendulum_open_loop(integrator="backward_euler", N=400):
    """run_iLQR_open_loop.py:16-69."""
    dyn = dict(kind="pendulum", dt=0.01, g=9.81, l=1.0, d=0.0, integrator=integrator)
    cost = dict(Q=np.eye(2), R=np.eye(1), Q_f=np.zeros((2, 2)), x_target=np.array([np.pi, 0.0]))
    return dict(dynamics=dyn, cost=cost, N=N, x0=np.array([1.0, 0.0]), U_init=np.zeros((1, N)),
                tol=1e-5, maxiter=100)


def pendulum_mpc(N=200):
    """run_iLQR_MPC.py:17-106: optimiser backward_euler, plant midpoint, maxiter 10."""
    dyn = dict(kind="pendulum", dt=0.01, g=9.81, l=1.0, d=0.0, integrator="backward_euler")
    cost = dict(Q=np.diag([10.0, 1.0]), R=np.eye(1), Q_f=np.diag([10.0, 10.0]), x_target=np.array([np.pi, 0.0]))
    return dict(dynamics=dyn, cost=cost, N=N, x0=np.zeros(2), U_init=np.zeros((1, N)), tol=1e-5, maxiter=10,
                plant_integrator="midpoint", n_sim=400)


def ua_double_pendulum(integrator="rk4", N=200):
    """run_iLQR_UA_MPC.py:17-67."""
    dyn = dict(kind="ua_double_pendulum", dt=0.01, g=9.81, m1=1.0, m2=1.0, l1=1.0, l2=1.0, d1=0.1, d2=0.1,
               theta1=1.0 / 12.0, theta2=1.0 / 12.0, integrator=integrator)
    cost = dict(Q=np.diag([5.0, 5.0, 0.1, 0.1]), R=np.diag([50.0]), Q_f=np.diag([1000.0, 1000.0, 10.0, 10.0]),
                x_target=np.array([np.pi, 0.0, 0.0, 0.0]))
    return dict(dynamics=dyn, cost=cost, N=N, x0=np.zeros(4), U_init=np.zeros((1, N)), tol=1e-5, maxiter=50,
                plant_integrator="backward_euler", n_sim=500)


def double_pendulum(integrator="rk4", N=100):
    """run_MPC_double_pendulum.py:17-63 (fully actuated, n_u = 2)."""
    dyn = dict(kind="double_pendulum", dt=0.01, g=9.81, m1=1.0, m2=1.0, l1=1.0, l2=1.0, d1=0.1, d2=0.1,
               theta1=1.0 / 12.0, theta2=1.0 / 12.0, integrator=integrator)
    cost = dict(Q=np.diag([5.0, 5.0, 0.1, 0.1]), R=np.diag([0.5, 0.5]), Q_f=np.diag([1000.0, 1000.0, 10.0, 10.0]),
                x_target=np.array([np.pi, 0.0, 0.0, 0.0]))
    return dict(dynamics=dyn, cost=cost, N=N, x0=np.array([0.0, 0.0, -10.0, 10.0]), U_init=np.zeros((2, N)),
                tol=1e-5, maxiter=50)


def ua_batch(batch, seed=0, restarts=False, N=200):
    """c2/c3/c4 inputs: x0_b = N(0, diag(.1,.1,.5,.5)^2) around the hanging equilibrium,
    U_init = 0 (or N(0, .1^2) for the random-restart variant)."""
    rng = np.random.default_rng(seed)
    x0 = rng.standard_normal((batch, 4)) * np.array([0.1, 0.1, 0.5, 0.5])
    if restarts:
        U = np.random.default_rng(seed + 1).standard_normal((batch, 1, N)) * 0.1
    else:
        U = np.zeros((batch, 1, N))
    return x0, U


def linear_quadratic(n=16, m=8, N=500, dt=0.01, seed=3):
    """c5: x+ = A x + B u with A = I + dt*G (spectral radius <= 1), B = dt*N(0,1), diagonal Q, R in
    U(.1, 1), Q_f = 10 Q; one (A, B, Q, R) shared by the whole batch."""
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((n, n)) / np.sqrt(n)
    A = np.eye(n) + dt * G
    rho = np.max(np.abs(np.linalg.eigvals(A)))
    if rho > 1.0:
        A = A / rho
    Bm = dt * rng.standard_normal((n, m))
    Q = np.diag(rng.uniform(0.1, 1.0, n))
    R = np.diag(rng.uniform(0.1, 1.0, m))
    dyn = dict(kind="linear", dt=dt, A=A, B=Bm, integrator="discrete")
    cost = dict(Q=Q, R=R, Q_f=10.0 * Q, x_target=np.zeros(n))
    return dict(dynamics=dyn, cost=cost, N=N, tol=1e-5, maxiter=20, seed=seed)


def lq_batch(batch, n, m, N, seed=3):
    rng = np.random.default_rng(seed + 100)
    return rng.standard_normal((batch, n)), np.zeros((batch, m, N))
