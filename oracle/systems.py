"""NumPy restatement of the reference's system layer (TEST INFRASTRUCTURE).

Follows /root/reference/python/class_files/systems/system_base.py (integrators
:50-74, backward-Euler quasi-Newton :88-140, implicit-function-theorem
Jacobians :146-188, which-derivative-is-which :203-219) and the three concrete
systems (pendulum_sys.py:60-98, UA_double_pendulum_sys.py:84-208,
double_pendulum_sys.py:84-206).

The reference obtains f_x, f_u, l_x, ... with JAX autodiff.  Here they are the
same mathematical objects written out analytically: the exact Jacobian of the
discrete map by the chain rule through the integrator stages (what
``jacfwd(self._f_fcn)`` evaluates, system_base.py:204-205) and the exact
gradients/Hessians of the quadratic costs (``grad``/``hessian``, :212-219).
tests/test_oracle_systems.py checks them against torch.func autodiff, central
differences and the MATLAB closed forms.

Everything is unbatched (one state, one control), exactly like the reference.
"""

from __future__ import annotations

import numpy as np

INTEGRATORS = ("euler", "midpoint", "rk4", "backward_euler", "discrete")


def _solve(A, b):
    """Dense solve in the dtype of the operands (LAPACK gesv, partial pivoting:
    the same factorisation ``jnp.linalg.solve`` uses)."""
    return np.linalg.solve(A, b)


class OracleSystem:
    """Counterpart of ``System`` (system_base.py:9-251).

    Subclasses provide ``f_cont``, ``f_cont_x``, ``f_cont_u`` (continuous
    dynamics and its Jacobians) -- the cost is the quadratic form every
    reference system uses.
    """

    def __init__(self, n_x, n_u, dt, x_target, Q, R, Q_f, integrator="rk4",
                 dtype=np.float64):
        if integrator not in INTEGRATORS:
            # system_base.py:198
            raise ValueError(
                f"Unknown integrator: '{integrator}'. Supported: 'rk4', "
                "'midpoint', 'euler', 'backward_euler'.")
        self.n_x, self.n_u = int(n_x), int(n_u)
        self.dtype = np.dtype(dtype)
        self.dt = self.dtype.type(dt)
        self.integrator = integrator
        c = lambda a, shp: np.asarray(a, dtype=self.dtype).reshape(shp)
        self.x_target = c(x_target, (n_x,))
        self.Q = c(Q, (n_x, n_x))
        self.R = c(R, (n_u, n_u))
        self.Q_f = c(Q_f, (n_x, n_x))
        # grad/hessian of 0.5*z'Pz is 0.5*(P+P')z and 0.5*(P+P')
        half = self.dtype.type(0.5)
        self._Qs = half * (self.Q + self.Q.T)
        self._Rs = half * (self.R + self.R.T)
        self._Qfs = half * (self.Q_f + self.Q_f.T)
        self._I = np.eye(n_x, dtype=self.dtype)

    # ---- continuous dynamics (subclass) ---------------------------------
    def f_cont(self, x, u):
        raise NotImplementedError

    def f_cont_x(self, x, u):
        raise NotImplementedError

    def f_cont_u(self, x, u):
        raise NotImplementedError

    # ---- cost: pendulum_sys.py:77-98, UA_double_pendulum_sys.py:114-136 --
    def l(self, x, u):
        dx = x - self.x_target
        half = self.dtype.type(0.5)
        return (half * dx @ self.Q @ dx + half * u @ self.R @ u) * self.dt

    def l_x(self, x, u):
        return (self._Qs @ (x - self.x_target)) * self.dt

    def l_u(self, x, u):
        return (self._Rs @ u) * self.dt

    def l_xx(self, x, u):
        return self._Qs * self.dt

    def l_uu(self, x, u):
        return self._Rs * self.dt

    def l_ux(self, x, u):
        # jacfwd(grad(l, 1), 0): shape (n_u, n_x)  (system_base.py:216)
        return np.zeros((self.n_u, self.n_x), dtype=self.dtype)

    def l_f(self, x):
        dx = x - self.x_target
        return self.dtype.type(0.5) * dx @ self.Q_f @ dx

    def l_f_x(self, x):
        return self._Qfs @ (x - self.x_target)

    def l_f_xx(self, x):
        return self._Qfs.copy()

    # ---- discrete dynamics ----------------------------------------------
    def f(self, x, u):
        x = np.asarray(x, dtype=self.dtype)
        u = np.asarray(u, dtype=self.dtype)
        dt = self.dt
        fc = self.f_cont
        two, six = self.dtype.type(2), self.dtype.type(6)
        if self.integrator == "discrete":
            return fc(x, u)
        if self.integrator == "euler":            # system_base.py:50-53
            return x + fc(x, u) * dt
        if self.integrator == "midpoint":         # :55-63
            k1 = fc(x, u)
            return x + dt * fc(x + (dt / two) * k1, u)
        if self.integrator == "rk4":              # :65-74
            k1 = fc(x, u)
            k2 = fc(x + dt / two * k1, u)
            k3 = fc(x + dt / two * k2, u)
            k4 = fc(x + dt * k3, u)
            return x + (dt / six) * (k1 + two * k2 + two * k3 + k4)
        return self._backward_euler(x, u)         # :88-140

    def _backward_euler(self, x, u):
        """Quasi-Newton backward Euler (system_base.py:88-140): explicit-Euler
        initial guess, ONE Jacobian I - dt*J_x at the guess reused for every
        iteration, stop when ||residual||_2 <= 1e-5 or after 20 iterations."""
        dt = self.dt
        res = lambda xn: xn - x - dt * self.f_cont(xn, u)
        xn = x + dt * self.f_cont(x, u)
        J = self._I - dt * self.f_cont_x(xn, u)
        F = res(xn)
        nrm = np.linalg.norm(F)
        k = 0
        tol = self.dtype.type(1e-5)
        while nrm > tol and k < 20:
            xn = xn + _solve(J, -F)
            F = res(xn)
            nrm = np.linalg.norm(F)
            k += 1
        return xn

    def f_x(self, x, u):
        return self._f_jac(x, u)[0]

    def f_u(self, x, u):
        return self._f_jac(x, u)[1]

    def _f_jac(self, x, u):
        """(df/dx, df/du) of the discrete map = forward-mode chain rule through
        the integrator stages (system_base.py:203-205) or, for backward Euler,
        the implicit-function theorem at the converged point (:146-188)."""
        x = np.asarray(x, dtype=self.dtype)
        u = np.asarray(u, dtype=self.dtype)
        dt, I = self.dt, self._I
        fc, Jx, Ju = self.f_cont, self.f_cont_x, self.f_cont_u
        two, six = self.dtype.type(2), self.dtype.type(6)
        it = self.integrator
        if it == "discrete":
            return Jx(x, u), Ju(x, u)
        if it == "euler":
            return I + dt * Jx(x, u), dt * Ju(x, u)
        if it == "midpoint":
            k1 = fc(x, u)
            xm = x + (dt / two) * k1
            Dx = I + (dt / two) * Jx(x, u)
            Du = (dt / two) * Ju(x, u)
            Jm = Jx(xm, u)
            return I + dt * (Jm @ Dx), dt * (Jm @ Du + Ju(xm, u))
        if it == "rk4":
            k1 = fc(x, u)
            K1x, K1u = Jx(x, u), Ju(x, u)
            x2 = x + dt / two * k1
            J2 = Jx(x2, u)
            K2x = J2 @ (I + dt / two * K1x)
            K2u = J2 @ (dt / two * K1u) + Ju(x2, u)
            k2 = fc(x2, u)
            x3 = x + dt / two * k2
            J3 = Jx(x3, u)
            K3x = J3 @ (I + dt / two * K2x)
            K3u = J3 @ (dt / two * K2u) + Ju(x3, u)
            k3 = fc(x3, u)
            x4 = x + dt * k3
            J4 = Jx(x4, u)
            K4x = J4 @ (I + dt * K3x)
            K4u = J4 @ (dt * K3u) + Ju(x4, u)
            fx = I + (dt / six) * (K1x + two * K2x + two * K3x + K4x)
            fu = (dt / six) * (K1u + two * K2u + two * K3u + K4u)
            return fx, fu
        # backward Euler, IFT (system_base.py:146-188)
        xs = self._backward_euler(x, u)
        J = I - dt * Jx(xs, u)
        return _solve(J, I), _solve(J, dt * Ju(xs, u))

    # the reference publishes these 12 names (system_base.py:223-251)
    f_fcn = f
    f_x_fcn = f_x
    f_u_fcn = f_u
    l_fcn = l
    l_x_fcn = l_x
    l_u_fcn = l_u
    l_xx_fcn = l_xx
    l_uu_fcn = l_uu
    l_ux_fcn = l_ux
    l_f_fcn = l_f
    l_f_x_fcn = l_f_x
    l_f_xx_fcn = l_f_xx


class PendulumOracle(OracleSystem):
    """pendulum_sys.py:12-98: x = [theta, theta_dot], u = [tau]."""

    def __init__(self, dt, x_target, Q, R, Q_f, g=9.81, l=1.0, d=0.01,
                 integrator="rk4", dtype=np.float64):
        super().__init__(2, 1, dt, x_target, Q, R, Q_f, integrator, dtype)
        t = self.dtype.type
        self.g, self.l_len, self.d = t(g), t(l), t(d)

    def f_cont(self, x, u):
        # pendulum_sys.py:60-75
        return np.array([x[1],
                         u[0] - self.d * x[1] - (self.g / self.l_len) * np.sin(x[0])],
                        dtype=self.dtype)

    def f_cont_x(self, x, u):
        z, o = self.dtype.type(0), self.dtype.type(1)
        return np.array([[z, o],
                         [-(self.g / self.l_len) * np.cos(x[0]), -self.d]],
                        dtype=self.dtype)

    def f_cont_u(self, x, u):
        return np.array([[0.0], [1.0]], dtype=self.dtype)


class _DoublePendulumBase(OracleSystem):
    """Shared physics of UA_double_pendulum_sys.py:84-208 and
    double_pendulum_sys.py:84-206: M(q) q_ddot = h(q, q_dot, tau),
    x = [q1, q2, q1_dot, q2_dot]."""

    n_act = None  # set by subclass

    def __init__(self, dt, x_target, Q, R, Q_f, g=9.81, m1=1.0, m2=1.0,
                 l1=1.0, l2=1.0, d1=0.01, d2=0.01, theta1=0.0, theta2=0.0,
                 integrator="rk4", dtype=np.float64):
        super().__init__(4, self.n_act, dt, x_target, Q, R, Q_f, integrator, dtype)
        t = self.dtype.type
        self.g, self.m1, self.m2 = t(g), t(m1), t(m2)
        self.l1, self.l2, self.d1, self.d2 = t(l1), t(l2), t(d1), t(d2)
        self.theta1, self.theta2 = t(theta1), t(theta2)

    def _mass(self, q2):
        # UA_double_pendulum_sys.py:140-162
        t = self.dtype.type
        m1, m2, l1, l2 = self.m1, self.m2, self.l1, self.l2
        c2 = np.cos(q2)
        m11 = (m1 * l1 ** 2) / t(4) + m2 * l1 ** 2 + (m2 * l2 ** 2) / t(4) \
            + m2 * l1 * l2 * c2 + self.theta1 + self.theta2
        m12 = (m2 * l2 ** 2) / t(4) + (m2 * l1 * l2 * c2) / t(2) + self.theta2
        m22 = (m2 * l2 ** 2) / t(4) + self.theta2
        return np.array([[m11, m12], [m12, m22]], dtype=self.dtype)

    def _act(self, u):
        raise NotImplementedError

    def _rhs(self, q, qd, u):
        # UA_double_pendulum_sys.py:164-208
        t = self.dtype.type
        m1, m2, l1, l2, g = self.m1, self.m2, self.l1, self.l2, self.g
        q1, q2 = q
        q1d, q2d = qd
        s1, s2, s12 = np.sin(q1), np.sin(q2), np.sin(q1 + q2)
        f_c = np.array([(m2 * l1 * l2 * s2 * (t(2) * q1d * q2d + q2d ** 2)) / t(2),
                        -(m2 * l1 * l2 * s2 * (q1d ** 2)) / t(2)], dtype=self.dtype)
        f_g = np.array([-m2 * g * (l2 * s12 / t(2) + l1 * s1) - (m1 * g * l1 * s1) / t(2),
                        -m2 * g * (l2 * s12) / t(2)], dtype=self.dtype)
        f_d = np.array([-self.d1 * q1d, -self.d2 * q2d], dtype=self.dtype)
        return self._act(u) + f_c + f_g + f_d

    def f_cont(self, x, u):
        # UA_double_pendulum_sys.py:84-111
        q, qd = x[:2], x[2:]
        qdd = _solve(self._mass(q[1]), self._rhs(q, qd, u))
        return np.concatenate([qd, qdd]).astype(self.dtype)

    def _qdd_jac(self, x, u):
        """d(q_ddot)/d[q1,q2,q1d,q2d] (2x4) and d(q_ddot)/du (2xn_u):
        M qdd = h  =>  d qdd = M^-1 (dh - dM qdd)."""
        t = self.dtype.type
        m1, m2, l1, l2, g = self.m1, self.m2, self.l1, self.l2, self.g
        q1, q2, q1d, q2d = x
        M = self._mass(q2)
        qdd = _solve(M, self._rhs(x[:2], x[2:], u))
        c1, c2, c12 = np.cos(q1), np.cos(q2), np.cos(q1 + q2)
        s2 = np.sin(q2)
        a = m2 * l1 * l2
        dh = np.zeros((2, 4), dtype=self.dtype)
        dh[0, 0] = -m2 * g * (l2 * c12 / t(2) + l1 * c1) - m1 * g * l1 * c1 / t(2)
        dh[1, 0] = -m2 * g * l2 * c12 / t(2)
        dh[0, 1] = a * c2 * (t(2) * q1d * q2d + q2d ** 2) / t(2) - m2 * g * l2 * c12 / t(2)
        dh[1, 1] = -a * c2 * q1d ** 2 / t(2) - m2 * g * l2 * c12 / t(2)
        dh[0, 2] = a * s2 * q2d - self.d1
        dh[1, 2] = -a * s2 * q1d
        dh[0, 3] = a * s2 * (q1d + q2d)
        dh[1, 3] = -self.d2
        dM = np.array([[-a * s2, -a * s2 / t(2)], [-a * s2 / t(2), t(0)]], dtype=self.dtype)
        dh[:, 1] -= dM @ qdd
        return _solve(M, dh), _solve(M, self._act_jac())

    def _act_jac(self):
        raise NotImplementedError

    def f_cont_x(self, x, u):
        J = np.zeros((4, 4), dtype=self.dtype)
        J[0, 2] = 1.0
        J[1, 3] = 1.0
        J[2:, :] = self._qdd_jac(x, u)[0]
        return J

    def f_cont_u(self, x, u):
        J = np.zeros((4, self.n_u), dtype=self.dtype)
        J[2:, :] = self._qdd_jac(x, u)[1]
        return J


class UADoublePendulumOracle(_DoublePendulumBase):
    """UA_double_pendulum_sys.py: torque on joint 1 only (f_act = [tau0, 0], :204)."""
    n_act = 1

    def _act(self, u):
        return np.array([u[0], 0.0], dtype=self.dtype)

    def _act_jac(self):
        return np.array([[1.0], [0.0]], dtype=self.dtype)


class DoublePendulumOracle(_DoublePendulumBase):
    """double_pendulum_sys.py: both joints actuated (f_act = [tau1, tau2], :202)."""
    n_act = 2

    def _act(self, u):
        return np.array([u[0], u[1]], dtype=self.dtype)

    def _act_jac(self):
        return np.eye(2, dtype=self.dtype)


class LinearQuadraticOracle(OracleSystem):
    """Linear dynamics x_dot = A x + B u (or, with integrator='discrete',
    x+ = A x + B u directly as in matlab/CLASSES/Linear_iLQR_CLASS.m:56-60)
    with the same quadratic cost scaled by dt (Linear_iLQR_CLASS.m:62-77)."""

    def __init__(self, dt, A, B, x_target, Q, R, Q_f, integrator="discrete",
                 dtype=np.float64):
        A = np.asarray(A)
        B = np.asarray(B)
        super().__init__(A.shape[0], B.shape[1], dt, x_target, Q, R, Q_f,
                         integrator, dtype)
        self.A = A.astype(self.dtype)
        self.B = B.astype(self.dtype)

    def f_cont(self, x, u):
        return self.A @ x + self.B @ u

    def f_cont_x(self, x, u):
        return self.A.copy()

    def f_cont_u(self, x, u):
        return self.B.copy()
