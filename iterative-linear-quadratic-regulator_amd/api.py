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


class RiccatiSweep:
    """The backward sweep as a stand-alone operator for callers that bring their own expansion (their autodiff, an
    identified or time-varying linear model): ``K, k = sweep(f_x, f_u, l_x, l_u, l_xx, l_ux, l_uu, V_x, V_xx)`` =
    the reverse scan of ``iLQR.backward_pass`` (iLQR_class.py:136-151) without the linearisation in front of it.

    Shapes (batch B, horizon N): f_x (B,N,n,n), f_u (B,N,n,m), l_x (B,N,n), l_u (B,N,m), l_xx (B,N,n,n),
    l_ux (B,N,m,n), l_uu (B,N,m,m), V_x (B,n), V_xx (B,n,n).  Returns K (B,N,m,n) and k = U_ff (B,m,N).

    The kernels exist for (n, m) in {(2,1), (4,1), (4,2), (8,4), (16,8)}; any n <= 16, m <= 8 is embedded in the
    next one by padding with states that have no dynamics and no cost and controls with unit curvature (l_uu = 1)
    and no effect -- their gains come out exactly zero and the original block's arithmetic is unchanged.
    """

    DIMS = ((2, 1), (4, 1), (4, 2), (8, 4), (16, 8))

    def __init__(self, n_x, n_u, N, batch, dtype=np.float64, mu=0.0, device=0):
        from . import _lib
        fit = [d for d in self.DIMS if d[0] >= n_x and d[1] >= n_u]
        if not fit:
            raise ValueError(f"no sweep kernel holds n_x = {n_x}, n_u = {n_u} (limits 16, 8)")
        self.n, self.m = int(n_x), int(n_u)
        self.np_, self.mp = fit[0]
        self.N, self.B, self.dtype = int(N), int(batch), np.dtype(dtype)
        n, m = self.np_, self.mp
        # the handle's system only fixes the dimensions: a linear system with neutral parameters
        params = np.zeros(n * n + n * m + n + 2 * n * n + m * m)
        self._h = _lib.Handle(system=_lib.SYS_LINEAR, n_x=n, n_u=m, horizon=self.N, batch=self.B, params=params,
                              dt=1.0, integrator="discrete", dtype=self.dtype, mu=mu, device=device)

    def __call__(self, f_x, f_u, l_x, l_u, l_xx, l_ux, l_uu, V_x, V_xx):
        B, N, n, m, P, Q = self.B, self.N, self.n, self.m, self.np_, self.mp
        dt = self.dtype

        def pad(a, shape, tail):
            a = np.asarray(a, dtype=dt).reshape((B,) + shape)
            out = np.zeros((B,) + tail, dtype=dt)
            out[tuple(slice(0, k) for k in a.shape)] = a
            return out

        fx, fu = pad(f_x, (N, n, n), (N, P, P)), pad(f_u, (N, n, m), (N, P, Q))
        lx, lu = pad(l_x, (N, n), (N, P)), pad(l_u, (N, m), (N, Q))
        lxx, lux, luu = pad(l_xx, (N, n, n), (N, P, P)), pad(l_ux, (N, m, n), (N, Q, P)), pad(l_uu, (N, m, m), (N, Q, Q))
        for j in range(m, Q):
            luu[:, :, j, j] = 1.0
        lin = np.concatenate([a.reshape(B, N, -1) for a in (fx, fu, lx, lu, lxx, lux, luu)], axis=2)
        term = np.concatenate([pad(V_x, (n,), (P,)), pad(V_xx, (n, n), (P, P)).reshape(B, -1)], axis=1)
        uff, K = self._h.backward_tensors(lin, term)
        return K[:, :, :m, :n], uff[:, :m, :]


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
