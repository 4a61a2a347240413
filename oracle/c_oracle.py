"""ctypes wrapper of the C oracle (oracle/c/ilqr_oracle.c) -- TEST INFRASTRUCTURE and the
``cpu_baseline`` of bench.py.  Same algorithm as oracle/ilqr.py, compiled; fp64 and fp32."""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libilqr_oracle.so")
_SYS = {"pendulum": 0, "ua_double_pendulum": 1, "double_pendulum": 2, "linear": 3}
_INT = {"euler": 0, "midpoint": 1, "rk4": 2, "backward_euler": 3, "discrete": 4}
_lib = None


def build():
    r = subprocess.run(["make", "-C", os.path.join(HERE, "c")], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("C oracle build failed:\n" + r.stdout + r.stderr)
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
    return _lib


def abi_params(dynamics, cost):
    """[system params | x_target | Q | R | Q_f] as float64 -- the block of include/ilqr_hip.h."""
    d = dict(dynamics)
    kind = d["kind"]
    if kind == "pendulum":
        sp = [d.get("g", 9.81), d.get("l", 1.0), d.get("d", 0.01)]
        n, m = 2, 1
    elif kind in ("ua_double_pendulum", "double_pendulum"):
        sp = [d.get("g", 9.81), d.get("m1", 1.0), d.get("m2", 1.0), d.get("l1", 1.0), d.get("l2", 1.0),
              d.get("d1", 0.01), d.get("d2", 0.01), d.get("theta1", 0.0), d.get("theta2", 0.0)]
        n, m = 4, (1 if kind == "ua_double_pendulum" else 2)
    else:
        A, B = np.asarray(d["A"], float), np.asarray(d["B"], float)
        sp = np.concatenate([A.ravel(), B.ravel()])
        n, m = A.shape[0], B.shape[1]
    p = np.concatenate([np.asarray(sp, float).ravel(), np.asarray(cost["x_target"], float).ravel(),
                        np.asarray(cost["Q"], float).ravel(), np.asarray(cost["R"], float).ravel(),
                        np.asarray(cost["Q_f"], float).ravel()])
    return p, len(np.asarray(sp).ravel()), n, m


class COracle:
    def __init__(self, dynamics, cost, dtype=np.float64, integrator=None):
        lib = load()
        self.dtype = np.dtype(dtype)
        sfx = "_f64" if self.dtype == np.float64 else "_f32"
        self.real = C.c_double if self.dtype == np.float64 else C.c_float
        p, nsys, n, m = abi_params(dynamics, cost)
        self.n, self.m = n, m
        integ = integrator or dynamics.get("integrator", "rk4")
        g = lambda name: getattr(lib, name + sfx)
        self._create, self._destroy = g("oracle_model_create"), g("oracle_model_destroy")
        self._backward, self._forward, self._solve, self._step = g("oracle_backward"), g("oracle_forward"), \
            g("oracle_solve_hist"), g("oracle_step")
        self._create.restype = C.c_void_p
        self._create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int]
        self._destroy.argtypes = [C.c_void_p]
        self._backward.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4
        self._forward.restype = self.real
        self._forward.argtypes = [C.c_void_p, C.c_int, C.c_void_p, self.real] + [C.c_void_p] * 6
        self._solve.restype = C.c_int
        self._solve.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_double, C.c_int, C.c_double, C.c_double,
                                                                          C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                                          C.c_void_p, C.c_void_p]
        self._step.argtypes = [C.c_void_p] * 6
        self.M = self._create(_SYS[dynamics["kind"]], _INT[integ], n, m, float(dynamics["dt"]),
                              p.ctypes.data_as(C.c_void_p), nsys)

    def __del__(self):
        try:
            self._destroy(self.M)
        except Exception:
            pass

    def _a(self, a):
        return np.ascontiguousarray(a, dtype=self.dtype)

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(C.c_void_p)

    def backward_pass(self, X, U):
        X, U = self._a(X), self._a(U)
        N = U.shape[1]
        Uff = np.zeros((self.m, N), self.dtype)
        K = np.zeros((N, self.m, self.n), self.dtype)
        self._backward(self.M, N, self._p(X), self._p(U), self._p(Uff), self._p(K))
        return Uff, K

    def forward_pass(self, x0, alpha, X, U, Uff, K):
        x0, X, U, Uff, K = map(self._a, (x0, X, U, Uff, K))
        N = U.shape[1]
        Xn, Un = np.zeros_like(X), np.zeros_like(U)
        c = self._forward(self.M, N, self._p(x0), alpha, self._p(X), self._p(U), self._p(Uff), self._p(K),
                          self._p(Xn), self._p(Un))
        return Xn, Un, self.dtype.type(c)

    def solve(self, x0, U_init, tol=1e-5, maxiter=100, alpha_factor=0.5, min_alpha=1e-8, n_trials=10,
              fixed_iters=0, state=None):
        """optimize_trajectory; ``state`` = (X, U_ff, K) carried from a previous solve (quirk Q1)."""
        U = self._a(U_init).copy()
        N = U.shape[1]
        x0 = self._a(x0)
        if state is None:
            X = np.zeros((self.n, N + 1), self.dtype)
            Uff = np.zeros((self.m, N), self.dtype)
            K = np.zeros((N, self.m, self.n), self.dtype)
        else:
            X, Uff, K = (self._a(s).copy() for s in state)
        cost = self.real(0)
        st = C.c_int(0)
        n_hist = max(int(fixed_iters), int(maxiter), 1)
        alpha_hist = np.zeros(n_hist, np.float64)
        cost_hist = np.zeros(n_hist, self.dtype)
        it = self._solve(self.M, N, self._p(x0), self._p(X), self._p(U), self._p(Uff), self._p(K), tol, maxiter,
                         alpha_factor, min_alpha, n_trials, fixed_iters, C.byref(cost), C.byref(st),
                         self._p(alpha_hist), self._p(cost_hist))
        status = {1: "converged", 2: "linesearch_failed", 3: "maxiter"}[st.value]
        # alphas / costs: accepted alpha (0 = none) and cost after each backward pass executed (iLQR_class.py:279-302)
        return dict(X=X, U=U, U_ff=Uff, K=K, cost=self.dtype.type(cost.value), iterations=it, status=status,
                    alphas=alpha_hist[:it].copy(), costs=cost_hist[:it].copy())

    def step(self, x, u, jac=True):
        x, u = self._a(x), self._a(u)
        xn = np.zeros(self.n, self.dtype)
        fx = np.zeros((self.n, self.n), self.dtype)
        fu = np.zeros((self.n, self.m), self.dtype)
        self._step(self.M, self._p(x), self._p(u), self._p(xn), self._p(fx) if jac else None,
                   self._p(fu) if jac else None)
        return xn, fx, fu
