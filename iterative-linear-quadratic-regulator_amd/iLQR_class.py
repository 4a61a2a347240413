"""Host-side mirror of the reference ``iLQR`` solver class, batched on the GPU.

Reference: python/class_files/iLQR_class.py:10-313.  Same constructor signature
(:18-27), same ``ValueError`` on a wrong ``U_init`` shape (:50-52), same public
state ``X, U, K, U_ff, x_0`` (:55-61) in the same layouts, same callables
``backward_pass(X, U)`` (:68) / ``forward_pass(x_0, alpha, X, U, U_ff, K)`` (:75) /
``optimize_trajectory()`` (:250-313), same printed messages.  All numerical work is
done by libilqr_hip.so (HIP kernels); this file only moves arguments across the
C-ABI.  There is no CPU fallback.

Batching (the build's extension): pass ``x_0`` of shape (B, n_x) and ``U_init`` of
shape (B, n_u, N) and every array gains a leading batch axis; B trajectories (MPC
instances, random restarts) are then solved as one job with every trial alpha of
the backtracking line search rolled out in parallel.
"""

from __future__ import annotations

import numpy as np

from . import _lib
from .systems.system_base import System, ready

STATUS_NAMES = {_lib.TRAJ_ACTIVE: "active", _lib.TRAJ_CONVERGED: "converged",
                _lib.TRAJ_LINESEARCH_FAILED: "linesearch_failed", _lib.TRAJ_MAXITER: "maxiter"}


def horizon_steps(T, dt):
    """N = len(arange(0, T + dt, dt)) - 1   (iLQR_class.py:46-47)."""
    return len(np.arange(0, T + dt, dt)) - 1


class iLQR:
    def __init__(self, system: System, T=None, x_0=None, U_init=None, tol=1e-5, maxiter=100,
                 alpha_factor=0.5, min_alpha=1e-8, verbose=True, *, N=None, n_alpha=None, n_trials=10,
                 dtype=None, device=0, mu=0.0, plant=None, flags=0, stream=None):
        self.system = system
        self.T = T
        self.tol, self.maxiter = tol, maxiter
        self.alpha_factor, self.min_alpha = alpha_factor, min_alpha
        self.verbose = verbose
        self.n_x, self.n_u, self.dt = system.n_x, system.n_u, system.dt
        if N is None:
            if T is None:
                raise ValueError("give the horizon either as T (seconds) or as N (steps)")
            self.tspan = np.arange(0, T + self.dt, self.dt)
            N = len(self.tspan) - 1
        else:
            self.tspan = np.arange(N + 1) * self.dt
        self.N = int(N)
        self.dtype = np.dtype(system.dtype if dtype is None else dtype)

        x_0 = np.asarray(x_0)
        U_init = np.asarray(U_init)
        self.batched = x_0.ndim == 2
        if self.batched:
            self.B = x_0.shape[0]
            expected = (self.B, self.n_u, self.N)
        else:
            self.B = 1
            expected = (self.n_u, self.N)
        if U_init.shape != expected:      # iLQR_class.py:50-52
            raise ValueError(f"U_init must have shape {expected}, but got {U_init.shape}")
        if x_0.shape[-1] != self.n_x:
            raise ValueError(f"x_0 must have {self.n_x} components, but got shape {x_0.shape}")

        if plant is not None and (type(plant) is not type(system) or not system.same_dynamics(plant)
                                  or plant.dt != system.dt):
            raise ValueError("the MPC plant must be the same system with the same parameters "
                             "(only its integrator may differ, run_iLQR_MPC.py:58-75)")
        self.plant = plant
        trial_count = 0
        a = 1.0
        for _ in range(n_trials):         # how many alphas the Python loop can reach (:281, :300-302)
            trial_count += 1
            a *= alpha_factor
            if a < min_alpha:
                break
        if n_alpha is None:
            n_alpha = min(trial_count, 16)
        self._h = system.make_handle(
            horizon=self.N, batch=self.B, dtype=self.dtype, n_alpha=n_alpha, n_trials=n_trials, tol=tol,
            maxiter=maxiter, alpha_factor=alpha_factor, min_alpha=min_alpha, mu=mu,
            plant_integrator=None if plant is None else plant.integrator, device=device, flags=flags,
            stream=stream)   # stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); None = a private one
        self._h.set_problem(x_0.reshape(self.B, self.n_x), U_init.reshape(self.B, self.n_u, self.N))
        self.status = None
        self.iterations = None

    # ---- state attributes (iLQR_class.py:55-61): reads are synchronised host copies ----
    def _out(self, a):
        return ready(a if self.batched else a[0])

    def _in(self, a, shape):
        a = np.asarray(a, dtype=self.dtype)
        return a if self.batched else a.reshape((1,) + tuple(shape))

    @property
    def X(self):
        return self._out(self._h.get(_lib.X))

    @X.setter
    def X(self, v):
        self._h.set(_lib.X, self._in(v, (self.n_x, self.N + 1)))

    @property
    def U(self):
        return self._out(self._h.get(_lib.U))

    @U.setter
    def U(self, v):
        self._h.set(_lib.U, self._in(v, (self.n_u, self.N)))

    @property
    def K(self):
        return self._out(self._h.get(_lib.K))

    @K.setter
    def K(self, v):
        self._h.set(_lib.K, self._in(v, (self.N, self.n_u, self.n_x)))

    @property
    def U_ff(self):
        return self._out(self._h.get(_lib.UFF))

    @U_ff.setter
    def U_ff(self, v):
        self._h.set(_lib.UFF, self._in(v, (self.n_u, self.N)))

    @property
    def x_0(self):
        return self._out(self._h.get(_lib.X0))

    @x_0.setter
    def x_0(self, v):
        self._h.set(_lib.X0, self._in(v, (self.n_x,)))

    @property
    def cost(self):
        c = self._h.get(_lib.COST)
        return c if self.batched else c[0]

    @property
    def handle(self):
        return self._h

    # ---- the two jitted passes of the reference, as pure functions --------------------------
    def backward_pass(self, X, U):
        """(U_ff, K) = backward sweep around (X, U)   (iLQR_class.py:122-161)."""
        uff, k = self._h.backward_pass(self._in(X, (self.n_x, self.N + 1)), self._in(U, (self.n_u, self.N)))
        return self._out(uff), self._out(k)

    def forward_pass(self, x_0, alpha, X, U, U_ff, K):
        """(X_new, U_new, cost) = rollout with u = U + alpha*U_ff + K (x - X)   (iLQR_class.py:193-247)."""
        Xn, Un, c = self._h.forward_pass(
            self._in(x_0, (self.n_x,)), float(alpha), self._in(X, (self.n_x, self.N + 1)),
            self._in(U, (self.n_u, self.N)), self._in(U_ff, (self.n_u, self.N)),
            self._in(K, (self.N, self.n_u, self.n_x)))
        return self._out(Xn), self._out(Un), (ready(c) if self.batched else c[0])

    # ---- optimize_trajectory (iLQR_class.py:250-313) --------------------------------------------
    def optimize_trajectory(self):
        h = self._h
        if self.verbose and not self.batched:
            self._solve_verbose()
        else:
            h.solve()
        st = h.get(_lib.STATUS)
        self.iterations = h.get(_lib.ITERS)
        self.status = [STATUS_NAMES[int(s) & 0xff] for s in st]
        self.non_pd = (st & _lib.TRAJ_FLAG_NON_PD) != 0
        if self.verbose and self.batched:
            counts = {v: self.status.count(v) for v in STATUS_NAMES.values()}
            print(f"iLQR batch of {self.B}: {counts}, iterations min/max "
                  f"{int(self.iterations.min())}/{int(self.iterations.max())}")
        cost = h.get(_lib.COST)
        if not self.batched:
            self.status, self.iterations = self.status[0], int(self.iterations[0])
        return self.X, self.U, (ready(cost) if self.batched else cost[0])

    def _solve_verbose(self):
        """Single trajectory with the reference's per-iteration printout: the same device
        stages, stepped one iteration at a time so the host can read the cost in between."""
        h = self._h
        h.initial_rollout()
        cost = h.get(_lib.COST)[0]
        print(f"Initial cost: {cost:.4f}")
        i = 0
        for i in range(self.maxiter):
            st = int(h.get(_lib.STATUS)[0]) & 0xff
            if st == _lib.TRAJ_CONVERGED:
                print(f"Converged at iteration {i}")
                break
            if st != _lib.TRAJ_ACTIVE:
                break
            h.iterate(1)
            alpha = h.get(_lib.ALPHA)[0]
            st = int(h.get(_lib.STATUS)[0]) & 0xff
            if st == _lib.TRAJ_LINESEARCH_FAILED:
                print(f"Warning: Line search failed at iteration {i+1}. Cost did not improve.")
                break
            cost = h.get(_lib.COST)[0]
            print(f"  Iter {i+1} (alpha={alpha:.2e}): Cost improved to {cost:.4f}")
        if i == self.maxiter - 1:
            print(f"Warning: Reached max iterations ({self.maxiter}) without converging.")

    # ---- MPC (run_iLQR_MPC.py:116-143), device-resident ---------------------------------------------
    def mpc_reset(self, x_0, U_init, keep_state=False):
        """Start the device-resident controller at plant state x_0 with warm start U_init.  keep_state=True keeps
        X, K, U_ff of the solve that ran before (the reference's run_iLQR_MPC.py enters its loop after a full warm-up
        optimize_trajectory(), :95); the default is a fresh solver (run_iLQR_UA_MPC.py:114-124)."""
        fn = self._h.mpc_rearm if keep_state else self._h.mpc_reset
        fn(self._in(x_0, (self.n_x,)), self._in(U_init, (self.n_u, self.N)))

    def mpc_run(self, n_steps):
        """n_steps receding-horizon steps on the device.  Returns (U_sim, X_sim, cost) with shapes
        (n_steps, [B,] n_u), (n_steps, [B,] n_x) (state AFTER each step) and (n_steps, [B])."""
        if self.plant is None:
            raise ValueError("construct the solver with plant=<System> to run MPC steps")
        u, x, c = self._h.mpc_run(n_steps)
        if not self.batched:
            u, x, c = u[:, 0], x[:, 0], c[:, 0]
        return ready(u), ready(x), ready(c)
