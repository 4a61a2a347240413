"""Linear system with the reference's quadratic cost: x_dot = A x + B u, or with
``integrator='discrete'`` the discrete map x+ = A x + B u itself.

Reference anchor: matlab/CLASSES/Linear_iLQR_CLASS.m:56-77 (f_fcn, l_fcn, l_f_fcn) and
matlab/functions/cont2disc.m:1-9; it is the template of the synthetic linear-quadratic
benchmark configuration.  Device code: csrc/dynamics.hpp ``Linear``.
"""
import numpy as np

from .. import _lib
from .system_base import System


class MyLinearSystem(System):
    SYSTEM_ID = _lib.SYS_LINEAR
    EXTRA_INTEGRATORS = ("discrete",)

    def __init__(self, dt, A, B, x_target, Q, R, Q_f, use_jit=True, integrator="discrete", dtype=np.float64):
        A = np.asarray(A, dtype=np.float64)
        B = np.asarray(B, dtype=np.float64)
        if A.ndim != 2 or A.shape[0] != A.shape[1] or B.ndim != 2 or B.shape[0] != A.shape[0]:
            raise ValueError(f"A must be (n, n) and B (n, m); got {A.shape} and {B.shape}")
        super().__init__(A.shape[0], B.shape[1], dt, use_jit=use_jit, integrator=integrator, dtype=dtype)
        self.A, self.B = A, B
        self._set_cost(x_target, Q, R, Q_f)

    def _system_params(self):
        return np.concatenate([self.A.ravel(), self.B.ravel()])
