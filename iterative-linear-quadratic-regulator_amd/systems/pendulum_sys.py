"""Simple pendulum, n_x = 2 ([theta, theta_dot]), n_u = 1 ([tau]).

Reference: python/class_files/systems/pendulum_sys.py:12-98 (constructor :22-55,
dynamics :60-75, costs :77-98).  Device code: csrc/dynamics.hpp ``Pendulum``.
"""
import numpy as np

from .. import _lib
from .system_base import System


class MyPendulum(System):
    SYSTEM_ID = _lib.SYS_PENDULUM

    def __init__(self, dt, x_target, Q, R, Q_f, g=9.81, l=1.0, d=0.01, use_jit=True, integrator="rk4",
                 dtype=np.float64):
        super().__init__(2, 1, dt, use_jit=use_jit, integrator=integrator, dtype=dtype)
        self.g, self.l, self.d = float(g), float(l), float(d)
        self._set_cost(x_target, Q, R, Q_f)

    def _system_params(self):
        return [self.g, self.l, self.d]
