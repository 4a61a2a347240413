"""Under-actuated double pendulum, n_x = 4 ([q1, q2, q1_dot, q2_dot]), n_u = 1 (torque on joint 1).

Reference: python/class_files/systems/UA_double_pendulum_sys.py:9-208 (constructor
:20-81, M(q) q_ddot = h :84-111, mass matrix :140-162, right-hand side :164-208).
Device code: csrc/dynamics.hpp ``DoublePendulum<T, 1>``.
"""
import numpy as np

from .. import _lib
from .system_base import System


class MyUADoublePendulum(System):
    SYSTEM_ID = _lib.SYS_UA_DOUBLE_PENDULUM
    N_U = 1

    def __init__(self, dt, x_target, Q, R, Q_f, g=9.81, m1=1.0, m2=1.0, l1=1.0, l2=1.0, d1=0.01, d2=0.01,
                 theta1=0.0, theta2=0.0, use_jit=True, integrator="rk4", dtype=np.float64):
        super().__init__(4, self.N_U, dt, use_jit=use_jit, integrator=integrator, dtype=dtype)
        self.g, self.m1, self.m2, self.l1, self.l2 = float(g), float(m1), float(m2), float(l1), float(l2)
        self.d1, self.d2, self.theta1, self.theta2 = float(d1), float(d2), float(theta1), float(theta2)
        self._set_cost(x_target, Q, R, Q_f)

    def _system_params(self):
        return [self.g, self.m1, self.m2, self.l1, self.l2, self.d1, self.d2, self.theta1, self.theta2]
