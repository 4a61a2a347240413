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


class DubinsCar(SymbolicSystem):
    """Constant-speed car; x = [px, py, heading], u = [turn rate] (n_x = 3, n_u = 1: rides the DPP sweep's 4 x 4 tile
    zero-padded)."""

    def __init__(self, dt, x_target, Q, R, Q_f, speed=1.0, **kw):
        self.speed = float(speed)
        super().__init__(3, 1, dt, x_target, Q, R, Q_f, **kw)

    def _f_cont_fcn(self, x, u):
        return [self.speed * sp.cos(x[2]), self.speed * sp.sin(x[2]), u[0]]


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


class SwingUpCartPole(CartPole):
    """Cart-pole with a non-quadratic user cost (the full subclass contract, system_base.py:255-275): pseudo-Huber
    angle error, a quartic control term, and a control/velocity cross term so that l_ux is not zero."""

    def __init__(self, dt, **kw):
        SymbolicSystem.__init__(self, 4, 1, dt, **kw)
        self.m_cart, self.m_pole, self.length, self.g = 1.0, 0.2, 0.5, 9.81

    def _l_fcn(self, x, u):
        p, th, pd, thd = x
        huber = sp.sqrt((th - sp.pi) ** 2 + 0.25)
        return self.dt * (0.5 * p ** 2 + 2.0 * huber + 0.05 * pd ** 2 + 0.05 * thd ** 2
                          + 0.01 * u[0] ** 2 + 0.001 * u[0] ** 4 + 0.004 * u[0] * pd)

    def _l_f_fcn(self, x):
        p, th, pd, thd = x
        return 50.0 * p ** 2 + 40.0 * (th - sp.pi) ** 2 + 20.0 * sp.sqrt((th - sp.pi) ** 2 + 0.25) + 5.0 * pd ** 2 \
            + 5.0 * thd ** 2


class ObstacleUnicycle(Unicycle):
    """Unicycle steering to a goal past a soft (Gaussian) obstacle; n_u = 2 with coupled controls.  The bump is
    kept weaker than the quadratic terms so the stage cost stays convex (plain iLQR, like the reference, has no
    safeguard against indefinite Hessians beyond the optional mu)."""

    goal = (1.0, 1.0)
    obstacle = (0.5, 0.4, 0.3, 0.03)   # centre x, y, width, height

    def __init__(self, dt, **kw):
        SymbolicSystem.__init__(self, 3, 2, dt, **kw)

    def _l_fcn(self, x, u):
        ox, oy, w, a = self.obstacle
        bump = a * sp.exp(-((x[0] - ox) ** 2 + (x[1] - oy) ** 2) / (2 * w ** 2))
        return self.dt * (0.5 * (x[0] - self.goal[0]) ** 2 + 0.5 * (x[1] - self.goal[1]) ** 2 + 0.05 * x[2] ** 2 + bump
                          + 0.1 * u[0] ** 2 + 0.1 * u[1] ** 2 + 0.05 * u[0] * u[1] + 0.02 * u[1] * x[2])

    def _l_f_fcn(self, x):
        return 50.0 * (x[0] - self.goal[0]) ** 2 + 50.0 * (x[1] - self.goal[1]) ** 2 + 2.0 * (x[2] - 0.5) ** 2


def example_problems(dtype=np.float64, integrator="rk4"):
    """name -> (system, N, x_0): the cases the tests and build() pre-compile."""
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
        "dubins": (DubinsCar(0.05, [2.0, 1.0, 0.0], np.diag([0.5, 1.0, 0.2]), [[0.2]], np.diag([5.0, 50.0, 10.0]), **kw),
                   60, np.array([0.0, 0.0, 0.2])),
        "quadrotor": (PlanarQuadrotor(0.02, [1.0, 1.0, 0, 0, 0, 0], np.diag([1.0, 1.0, 1.0, 0.1, 0.1, 0.1]),
                                      np.diag([0.1, 0.1]), np.diag([100.0, 100.0, 10.0, 10.0, 10.0, 1.0]), **kw),
                      50, np.array([0.0, 0.0, 0.0, 0, 0, 0])),
        "swingup_cartpole": (SwingUpCartPole(0.02, **kw), 80, np.array([0.0, 0.3, 0, 0])),
        "obstacle_unicycle": (ObstacleUnicycle(0.05, **kw), 60, np.array([0.0, 0.0, 0.3])),
    }


def bench_cases():
    """User systems at the north-star shape (B = 4096, N = 200, n = 4, m = 1) for tools/bench_configs.py: the symbolic
    restatement of the reference's UA double pendulum (same parameters as problems.ua_double_pendulum) and the
    cart-pole with its non-quadratic user cost.  name -> (system, N, B, x_0 centre)."""
    from .. import problems
    p = problems.ua_double_pendulum(N=200)
    d, c = p["dynamics"], p["cost"]
    phys = {k: d[k] for k in ("g", "m1", "m2", "l1", "l2", "d1", "d2") if k in d}
    out = {}
    for tag, dt in (("f32", np.float32), ("f64", np.float64)):
        sym = SymbolicUADoublePendulum(d["dt"], c["x_target"], c["Q"], c["R"], c["Q_f"], integrator=d["integrator"],
                                       dtype=dt, **phys)
        out[f"user {tag}: symbolic UA double pendulum rk4"] = (sym, 200, 4096, np.asarray(p["x0"], dtype=np.float64))
    out["user float32: cart-pole + non-quadratic cost rk4"] = (SwingUpCartPole(0.02, dtype=np.float32), 200, 4096,
                                                              np.array([0.0, 0.3, 0.0, 0.0]))
    return out
