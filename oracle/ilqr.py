"""NumPy restatement of the reference iLQR solver (TEST INFRASTRUCTURE).

Follows /root/reference/python/class_files/iLQR_class.py: backward step :79-119,
backward sweep + terminal condition :122-161, rollout step :164-190, rollout
:193-247, outer loop / backtracking line search / acceptance :250-313, and the
MPC loop of /root/reference/python/run_iLQR_MPC.py:116-143.

The reference runs these as ``jax.lax.scan`` bodies; here they are explicit
per-timestep loops (the form of matlab/CLASSES/iLQR_CLASS.m:106-143).  Layouts
are the reference's: X (n_x, N+1), U (n_u, N), U_ff (n_u, N),
K (N, n_u, n_x).
"""

from __future__ import annotations

import numpy as np


def horizon_steps(T, dt):
    """N = len(arange(0, T+dt, dt)) - 1   (iLQR_class.py:46-47)."""
    return len(np.arange(0, T + dt, dt)) - 1


def backward_step(sys, x, u, V_x, V_xx, mu=0.0):
    """One Riccati step (iLQR_class.py:79-119).  ``mu`` is the build's
    Levenberg extension (Q_uu + mu*I); mu = 0 is the reference."""
    l_x, l_u = sys.l_x(x, u), sys.l_u(x, u)
    l_xx, l_ux, l_uu = sys.l_xx(x, u), sys.l_ux(x, u), sys.l_uu(x, u)
    f_x, f_u = sys.f_x(x, u), sys.f_u(x, u)
    # :100-104
    Q_x = l_x + f_x.T @ V_x
    Q_u = l_u + f_u.T @ V_x
    Q_xx = l_xx + f_x.T @ V_xx @ f_x
    Q_ux = l_ux + f_u.T @ V_xx @ f_x
    Q_uu = l_uu + f_u.T @ V_xx @ f_u
    if mu:
        Q_uu_reg = Q_uu + sys.dtype.type(mu) * np.eye(sys.n_u, dtype=sys.dtype)
    else:
        Q_uu_reg = Q_uu
    # :109-110
    K = -np.linalg.solve(Q_uu_reg, Q_ux)
    k = -np.linalg.solve(Q_uu_reg, Q_u)
    if mu:
        # full (Joseph-style) value update, exact for any gain
        V_x_prev = Q_x + K.T @ (Q_uu @ k) + K.T @ Q_u + Q_ux.T @ k
        V_xx_prev = Q_xx + K.T @ Q_uu @ K + K.T @ Q_ux + Q_ux.T @ K
    else:
        # :113-114 (short form)
        V_x_prev = Q_x + K.T @ Q_u
        V_xx_prev = Q_xx + Q_ux.T @ K
    return K, k, V_x_prev, V_xx_prev


def backward_pass(sys, X, U, mu=0.0, return_value=False):
    """Full backward sweep (iLQR_class.py:122-161).

    X (n_x, N+1), U (n_u, N)  ->  U_ff (n_u, N), K (N, n_u, n_x)."""
    X = np.asarray(X, dtype=sys.dtype)
    U = np.asarray(U, dtype=sys.dtype)
    N = U.shape[1]
    x_N = X[:, -1]
    V_x, V_xx = sys.l_f_x(x_N), sys.l_f_xx(x_N)      # :136-138
    U_ff = np.zeros((sys.n_u, N), dtype=sys.dtype)
    K = np.zeros((N, sys.n_u, sys.n_x), dtype=sys.dtype)
    for t in range(N - 1, -1, -1):                    # reverse scan :149-151
        K[t], U_ff[:, t], V_x, V_xx = backward_step(sys, X[:, t], U[:, t], V_x, V_xx, mu)
    if return_value:
        return U_ff, K, V_x, V_xx
    return U_ff, K


class _Record:
    """One expansion record seen through the System interface backward_step reads."""

    def __init__(self, rec, dtype):
        self._r, self.dtype = rec, np.dtype(dtype)
        self.n_u = rec["l_u"].shape[0]
        for name in ("f_x", "f_u", "l_x", "l_u", "l_xx", "l_ux", "l_uu"):
            setattr(self, name, (lambda v: (lambda x, u: v))(np.asarray(rec[name], dtype=dtype)))


def backward_tensors(f_x, f_u, l_x, l_u, l_xx, l_ux, l_uu, V_x, V_xx, mu=0.0, dtype=np.float64):
    """The reverse scan of iLQR_class.py:136-151 on a given expansion (one trajectory):
    f_x (N,n,n), f_u (N,n,m), l_x (N,n), l_u (N,m), l_xx (N,n,n), l_ux (N,m,n), l_uu (N,m,m), V_x (n), V_xx (n,n)
    -> U_ff (m,N), K (N,m,n).  Same step function as backward_pass."""
    N, n, m = f_u.shape
    V_x, V_xx = np.asarray(V_x, dtype=dtype), np.asarray(V_xx, dtype=dtype)
    U_ff, K = np.zeros((m, N), dtype=dtype), np.zeros((N, m, n), dtype=dtype)
    for t in range(N - 1, -1, -1):
        rec = _Record(dict(f_x=f_x[t], f_u=f_u[t], l_x=l_x[t], l_u=l_u[t], l_xx=l_xx[t], l_ux=l_ux[t], l_uu=l_uu[t]), dtype)
        K[t], U_ff[:, t], V_x, V_xx = backward_step(rec, None, None, V_x, V_xx, mu)
    return U_ff, K


def forward_pass(sys, x_0, alpha, X_old, U_old, U_ff, K):
    """Rollout with the affine control law (iLQR_class.py:164-247).

    Returns X_new (n_x, N+1), U_new (n_u, N), cost."""
    dt = sys.dtype
    X_old = np.asarray(X_old, dtype=dt)
    U_old = np.asarray(U_old, dtype=dt)
    U_ff = np.asarray(U_ff, dtype=dt)
    K = np.asarray(K, dtype=dt)
    alpha = dt.type(alpha)
    N = U_old.shape[1]
    X_new = np.zeros((sys.n_x, N + 1), dtype=dt)
    U_new = np.zeros((sys.n_u, N), dtype=dt)
    x = np.asarray(x_0, dtype=dt).copy()
    cost = dt.type(0.0)                               # :216
    for t in range(N):
        u = U_old[:, t] + alpha * U_ff[:, t] + K[t] @ (x - X_old[:, t])   # :181-182
        X_new[:, t], U_new[:, t] = x, u               # :188 (state/control *used*)
        cost = cost + sys.l(x, u)                     # :187, :340
        x = sys.f(x, u)                               # :185, :339
    X_new[:, N] = x                                   # :241
    cost = cost + sys.l_f(x)                          # :245
    return X_new, U_new, cost


class iLQROracle:
    """Stateful counterpart of ``iLQR`` (iLQR_class.py:10-313), including the
    state carried between solves."""

    def __init__(self, system, T=None, x_0=None, U_init=None, tol=1e-5,
                 maxiter=100, alpha_factor=0.5, min_alpha=1e-8, verbose=False,
                 N=None, mu=0.0, n_trials=10):
        self.system = system
        dt = system.dtype
        self.N = int(N) if N is not None else horizon_steps(T, float(system.dt))
        self.n_x, self.n_u = system.n_x, system.n_u
        self.x_0 = np.asarray(x_0, dtype=dt)
        U_init = np.asarray(U_init)
        if U_init.shape != (self.n_u, self.N):        # :50-52
            raise ValueError(f"U_init must have shape {(self.n_u, self.N)}, "
                             f"but got {U_init.shape}")
        self.tol, self.maxiter = tol, maxiter
        self.alpha_factor, self.min_alpha = alpha_factor, min_alpha
        self.verbose, self.mu, self.n_trials = verbose, mu, n_trials
        self.X = np.zeros((self.n_x, self.N + 1), dtype=dt)     # :55
        self.U = U_init.astype(dt)
        self.K = np.zeros((self.N, self.n_u, self.n_x), dtype=dt)
        self.U_ff = np.zeros((self.n_u, self.N), dtype=dt)
        self.history = []       # (iteration, alpha, cost) of accepted steps
        self.status = None      # 'converged' | 'linesearch_failed' | 'maxiter'
        self.iterations = 0     # backward passes executed in the last solve

    def backward_pass(self, X, U):
        return backward_pass(self.system, X, U, self.mu)

    def forward_pass(self, x_0, alpha, X_old, U_old, U_ff, K):
        return forward_pass(self.system, x_0, alpha, X_old, U_old, U_ff, K)

    def optimize_trajectory(self):
        """iLQR_class.py:250-313."""
        self.history = []
        self.X, self.U, cost = self.forward_pass(        # :257-259, alpha = 0
            self.x_0, 0.0, self.X, self.U, self.U_ff, self.K)
        self.initial_cost = cost
        cost_prev = cost
        self.status = "maxiter"
        self.iterations = 0
        for i in range(self.maxiter):
            if i > 0 and abs(cost - cost_prev) <= self.tol:      # :267
                self.status = "converged"
                break
            cost_prev = cost
            self.U_ff, self.K = self.backward_pass(self.X, self.U)   # :275
            self.iterations += 1
            alpha = 1.0
            accepted = False
            for _ in range(self.n_trials):                        # :281
                X_new, U_new, cost_new = self.forward_pass(
                    self.x_0, alpha, self.X, self.U, self.U_ff, self.K)
                if cost_new <= cost:                              # :289
                    self.X, self.U, cost = X_new, U_new, cost_new
                    accepted = True
                    self.history.append((i + 1, alpha, cost))
                    break
                alpha *= self.alpha_factor                        # :300
                if alpha < self.min_alpha:                        # :301
                    break
            if not accepted:                                      # :304-307
                self.status = "linesearch_failed"
                break
        return self.X, self.U, cost


def mpc_closed_loop(solver, plant, x_0, U_init, n_sim, warmup=False):
    """Receding-horizon loop of run_iLQR_MPC.py:116-143: set x_0 and U, solve,
    apply the first control to the plant, shift the warm start (repeat last).

    warmup=True reproduces run_iLQR_MPC.py:95: ONE full optimize_trajectory() on the same solver object before
    the loop ("JIT warm-up").  It leaves X, K, U_ff behind, and the loop's first solve starts from them (its alpha = 0
    rollout is u = U_init + K_warm (x - X_warm), iLQR_class.py:257-259).  run_iLQR_UA_MPC.py warms
    up through the pure backward_pass / forward_pass instead (:114-124) and starts cold: warmup=False."""
    if warmup:
        solver.optimize_trajectory()
    dt = solver.system.dtype
    X_sim = np.zeros((solver.n_x, n_sim + 1), dtype=dt)
    U_sim = np.zeros((solver.n_u, n_sim), dtype=dt)
    costs = np.zeros(n_sim, dtype=dt)
    x = np.asarray(x_0, dtype=dt)
    X_sim[:, 0] = x
    U_guess = np.asarray(U_init, dtype=dt)
    for k in range(n_sim):
        solver.x_0 = x                                # :118
        solver.U = U_guess                            # :121
        _, U_bar, cost = solver.optimize_trajectory() # :124
        u0 = U_bar[:, 0]                              # :127
        x = plant.f(x, u0)                            # :130
        U_sim[:, k], X_sim[:, k + 1], costs[k] = u0, x, cost
        U_guess = np.concatenate([U_bar[:, 1:], U_bar[:, -1:]], axis=1)   # :137
    return X_sim, U_sim, costs
