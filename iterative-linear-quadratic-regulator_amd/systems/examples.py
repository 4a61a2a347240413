"""Example user-defined systems (what a user of the reference would write as a ``System`` subclass,
system_base.py:255-275), used by the docs, the GPU tests and scripts/run_iLQR_cartpole.py.

Each one only states its continuous dynamics with sympy math; see custom_sys.py for what happens next.
"""
import numpy as np
import sympy as sp

from .custom_sys import SymbolicSystem


class SymbolicPendulum(SymbolicSystem):
    """The reference pendulum (pendulum_sys.py:60-75) re-stated as a user system: must agree with the
    built-in ``MyPendulum`` to rounding."""

    def __init__(self, dt, x_target, Q, R, Q_f, g=9.81, l=1.0, d=0.01, **kw):
        self.g, self.l, self.d = float(g), float(l), float(d)
        super().__init__(2, 1, dt, x_target, Q, R, Q_f, **kw)

    def _f_cont_fcn(self, x, u):
        theta, theta_dot = x
        return [theta_dot, u[0] - self.d * theta_dot - self.g / self.l * sp.sin(theta)]


class SymbolicUADoublePendulum(SymbolicSystem):
    """The reference's underactuated double pendulum (UA_double_pendulum_sys.py:84-112: M(q) qdd = h(q, qd, u))
    with the 2x2 solve written out; n_x = 4, n_u = 1, so it runs on the DPP backward sweep."""

    def __init__(self, dt, x_target, Q, R, Q_f, g=9.81, m1=1.0, m2=1.0, l1=1.0, l2=1.0, d1=0.01, d2=0.01, **kw):
        self.g, self.m1, self.m2, self.l1, self.l2, self.d1, self.d2 = map(float, (g, m1, m2, l1, l2, d1, d2))
        self.theta1 = self.m1 * self.l1 ** 2 / 12.0
        self.theta2 = self.m2 * self.l2 ** 2 / 12.0
        super().__init__(4, 1, dt, x_target, Q, R, Q_f, **kw)

    def _f_cont_fcn(self, x, u):
        q1, q2, q1d, q2d = x
        m1, m2, l1, l2, g = self.m1, self.m2, self.l1, self.l2, self.g
        m11 = m1 * l1 ** 2 / 4 + m2 * (l1 ** 2 + l2 ** 2 / 4 + l1 * l2 * sp.cos(q2)) + self.theta1 + self.theta2
        m12 = m2 * (l2 ** 2 / 4 + l1 * l2 * sp.cos(q2) / 2) + self.theta2
        m22 = m2 * l2 ** 2 / 4 + self.theta2
        h1 = (u[0] + m2 * l1 * l2 * sp.sin(q2) * (2 * q1d * q2d + q2d ** 2) / 2 - m2 * g * l2 * sp.sin(q1 + q2) / 2
              - (m2 + m1 / 2) * g * l1 * sp.sin(q1) - self.d1 * q1d)
        h2 = -m2 * l1 * l2 * sp.sin(q2) * q1d ** 2 / 2 - m2 * g * l2 * sp.sin(q1 + q2) / 2 - self.d2 * q2d
        det = m11 * m22 - m12 * m12
        return [q1d, q2d, (m22 * h1 - m12 * h2) / det, (m11 * h2 - m12 * h1) / det]


class CartPole(SymbolicSystem):
    """Cart with a point-mass pole; x = [p, theta, p_dot, theta_dot] (theta = 0 hanging down), u = [force]."""

    def __init__(self, dt, x_target, Q, R, Q_f, m_cart=1.0, m_pole=0.2, length=0.5, g=9.81, **kw):
        self.m_cart, self.m_pole, self.length, self.g = float(m_cart), float(m_pole), float(length), float(g)
        super().__init__(4, 1, dt, x_target, Q, R, Q_f, **kw)

    def _f_cont_fcn(self, x, u):
        _, th, pd, thd = x
        mc, mp, l, g = self.m_cart, self.m_pole, self.length, self.g
        s, c = sp.sin(th), sp.cos(th)
        den = mc + mp * s ** 2
        pdd = (u[0] + mp * s * (l * thd ** 2 + g * c)) / den
        thdd = (-u[0] * c - mp * l * thd ** 2 * c * s - (mc + mp) * g * s) / (l * den)
        return [pd, thd, pdd, thdd]


class Unicycle(SymbolicSystem):
    """Kinematic unicycle; x = [px, py, heading], u = [speed, turn rate] (n_x = 3, n_u = 2)."""

    def __init__(self, dt, x_target, Q, R, Q_f, **kw):
        super().__init__(3, 2, dt, x_target, Q, R, Q_f, **kw)

    def _f_cont_fcn(self, x, u):
        return [u[0] * sp.cos(x[2]), u[0] * sp.sin(x[2]), u[1]]


class PlanarQuadrotor(SymbolicSystem):
    """Planar quadrotor; x = [px, pz, phi, vx, vz, phi_dot], u = [thrust_left, thrust_right] (n_x = 6, n_u = 2)."""

    def __init__(self, dt, x_target, Q, R, Q_f, mass=0.5, inertia=0.01, arm=0.2, g=9.81, **kw):
        self.mass, self.inertia, self.arm, self.g = float(mass), float(inertia), float(arm), float(g)
        super().__init__(6, 2, dt, x_target, Q, R, Q_f, **kw)

    def _f_cont_fcn(self, x, u):
        _, _, phi, vx, vz, phid = x
        thrust = u[0] + u[1]
        return [vx, vz, phid, -thrust * sp.sin(phi) / self.mass, thrust * sp.cos(phi) / self.mass - self.g,
                self.arm * (u[1] - u[0]) / self.inertia]


def example_problems(dtype=np.float64, integrator="rk4"):
    """name -> (system, N, x_0, initial control scale): the cases the tests and build() pre-compile."""
    kw = dict(dtype=dtype, integrator=integrator)
    pi = np.pi
    return {
        "sym_pendulum": (SymbolicPendulum(0.02, [pi, 0.0], np.diag([0.1, 0.01]), [[0.01]], np.diag([100.0, 10.0]), **kw),
                         100, np.array([0.0, 0.0])),
        "sym_ua": (SymbolicUADoublePendulum(0.01, [pi, 0, 0, 0], np.diag([1.0, 1.0, 0.1, 0.1]), [[0.01]],
                                           np.diag([100.0, 100.0, 10.0, 10.0]), **kw), 60, np.array([0.3, -0.2, 0, 0])),
        "cartpole": (CartPole(0.02, [0, pi, 0, 0], np.diag([1.0, 1.0, 0.1, 0.1]), [[0.01]],
                              np.diag([100.0, 100.0, 10.0, 10.0]), **kw), 80, np.array([0.0, 0.0, 0, 0])),
        "unicycle": (Unicycle(0.05, [1.0, 1.0, 0.5 * pi], np.diag([1.0, 1.0, 0.1]), np.diag([0.1, 0.1]),
                              np.diag([50.0, 50.0, 5.0]), **kw), 60, np.array([0.0, 0.0, 0.0])),
        "quadrotor": (PlanarQuadrotor(0.02, [1.0, 1.0, 0, 0, 0, 0], np.diag([1.0, 1.0, 1.0, 0.1, 0.1, 0.1]),
                                      np.diag([0.1, 0.1]), np.diag([100.0, 100.0, 10.0, 10.0, 10.0, 1.0]), **kw),
                      50, np.array([0.0, 0.0, 0.0, 0, 0, 0])),
    }
