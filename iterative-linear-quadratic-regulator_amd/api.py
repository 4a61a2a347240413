"""Functional facade named by the north star: ``solve(dynamics, cost, x0, U_init)``
and an MPC step, on top of the same ``iLQR`` object the drivers use.

``dynamics`` is a dict describing one of the built-in systems, e.g.
``{"kind": "ua_double_pendulum", "dt": 0.01, "integrator": "rk4", "m1": 1.0, ...}``
(keys as the reference constructors' keyword arguments, UA_double_pendulum_sys.py:20-38),
or an already constructed ``System``.  ``cost`` is a dict with ``Q, R, Q_f, x_target``
(the quadratic cost every reference system uses, pendulum_sys.py:77-98).
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .iLQR_class import iLQR
from .systems import (System, MyPendulum, MyUADoublePendulum, MyDoublePendulum, MyLinearSystem)

_KINDS = {"pendulum": MyPendulum, "ua_double_pendulum": MyUADoublePendulum,
          "double_pendulum": MyDoublePendulum, "linear": MyLinearSystem}


def make_system(dynamics, cost=None, dtype=np.float64):
    if isinstance(dynamics, System):
        return dynamics
    d = dict(dynamics)
    kind = d.pop("kind")
    if kind not in _KINDS:
        raise ValueError(f"unknown system kind '{kind}'; known: {sorted(_KINDS)}")
    if cost is None:
        raise ValueError("cost = {Q, R, Q_f, x_target} is required with a dynamics description")
    return _KINDS[kind](x_target=cost["x_target"], Q=cost["Q"], R=cost["R"], Q_f=cost["Q_f"], dtype=dtype, **d)


@dataclass
class SolveResult:
    X: np.ndarray        # ([B,] n_x, N+1)
    U: np.ndarray        # ([B,] n_u, N)
    cost: np.ndarray     # ([B])
    K: np.ndarray        # ([B,] N, n_u, n_x)
    k: np.ndarray        # ([B,] n_u, N)   feed-forward U_ff
    iters: np.ndarray    # ([B]) backward passes executed
    status: object       # 'converged' | 'linesearch_failed' | 'maxiter' (list for a batch)
    solver: iLQR


def solve(dynamics, cost, x0, U_init, *, T=None, N=None, tol=1e-5, maxiter=100, alpha_factor=0.5,
          min_alpha=1e-8, n_alpha=None, mu=0.0, dtype=np.float64, device=0, verbose=False):
    """Solve one trajectory (x0 (n,), U_init (m, N)) or a batch (x0 (B, n), U_init (B, m, N))."""
    system = make_system(dynamics, cost, dtype)
    U_init = np.asarray(U_init)
    if N is None and T is None:
        N = U_init.shape[-1]
    s = iLQR(system, T, x0, U_init, tol=tol, maxiter=maxiter, alpha_factor=alpha_factor, min_alpha=min_alpha,
             verbose=verbose, N=N, n_alpha=n_alpha, mu=mu, dtype=dtype, device=device)
    X, U, c = s.optimize_trajectory()
    return SolveResult(X=X, U=U, cost=c, K=s.K, k=s.U_ff, iters=s.iterations, status=s.status, solver=s)


@dataclass
class MPCState:
    solver: iLQR
    steps_done: int = 0


def mpc_init(dynamics, cost, x0, U_init, *, plant_integrator="midpoint", T=None, N=None, tol=1e-5, maxiter=10,
             n_alpha=None, dtype=np.float64, device=0):
    """Receding-horizon controller state (run_iLQR_MPC.py:58-106): optimiser model = ``dynamics``,
    plant = the same system with ``plant_integrator``."""
    system = make_system(dynamics, cost, dtype)
    if isinstance(dynamics, System):
        raise ValueError("mpc_init needs a dynamics description (dict) so it can build the plant twin")
    plant = make_system({**dict(dynamics), "integrator": plant_integrator}, cost, dtype)
    U_init = np.asarray(U_init)
    if N is None and T is None:
        N = U_init.shape[-1]
    s = iLQR(system, T, x0, U_init, tol=tol, maxiter=maxiter, verbose=False, N=N, n_alpha=n_alpha, dtype=dtype,
             device=device, plant=plant)
    s.mpc_reset(x0, U_init)
    return MPCState(solver=s)


def mpc_step(state: MPCState, x_now=None):
    """One MPC step (run_iLQR_MPC.py:116-143).  With ``x_now`` the plant state is overwritten by the
    caller's measurement first; otherwise the internal plant model supplies it.
    Returns (u0, state) -- u0 ([B,] n_u) is the control applied at this step."""
    s = state.solver
    if x_now is not None:
        x_now = s._in(x_now, (s.n_x,))
        s._h.set(_lib_field("PLANT_X"), x_now)
        s._h.set(_lib_field("X0"), x_now)
    u, _, _ = s.mpc_run(1)
    state.steps_done += 1
    return u[0], state


def _lib_field(name):
    from . import _lib
    return getattr(_lib, name)
