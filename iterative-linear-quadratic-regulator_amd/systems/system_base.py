"""Host-side mirror of the reference ``System`` base class.

Reference: python/class_files/systems/system_base.py:9-251.  There the class
turns three JAX methods into twelve jitted callables by autodiff; here a system
is a *description* (system id + parameter block + integrator) of dynamics whose
device implementation is compiled into libilqr_hip.so, and the twelve public
callables (``f_fcn`` ... ``l_f_xx_fcn``, system_base.py:223-251) evaluate on the
GPU through ``ilqr_eval_points``.  Same names, same argument meaning, same
shapes, same ``ValueError`` for an unknown integrator (system_base.py:198).
"""

from __future__ import annotations

import numpy as np

from .. import _lib

SUPPORTED_INTEGRATORS = ("rk4", "midpoint", "euler", "backward_euler")


class _ReadyArray(np.ndarray):
    """ndarray whose ``block_until_ready()`` is a no-op, so driver code written for
    JAX arrays (run_iLQR_open_loop.py:83,93) keeps working on results that are already
    synchronised host copies."""

    def block_until_ready(self):
        return self


def ready(a):
    return np.asarray(a).view(_ReadyArray)


class System:
    """Base of the built-in systems.

    Subclasses set ``SYSTEM_ID`` and implement ``_system_params()`` (the leading part
    of the C-ABI parameter block, include/ilqr_hip.h).  ``use_jit`` is accepted for
    signature compatibility and ignored (everything is pre-compiled device code).
    """

    SYSTEM_ID = None
    EXTRA_INTEGRATORS = ()

    def __init__(self, n_x, n_u, dt, use_jit=True, integrator="rk4", dtype=np.float64):
        self.n_x = int(n_x)
        self.n_u = int(n_u)
        self.dt = float(dt)
        self.use_jit = use_jit
        if integrator not in SUPPORTED_INTEGRATORS + tuple(self.EXTRA_INTEGRATORS):
            raise ValueError(f"Unknown integrator: '{integrator}'. Supported: 'rk4', 'midpoint', 'euler', "
                             "'backward_euler'.")
        self.integrator = integrator
        self.dtype = np.dtype(dtype)
        _lib.np_dtype(self.dtype)  # validates
        self._probe = None

    # ---- description handed to the C-ABI ----------------------------------------------
    def _system_params(self):
        raise NotImplementedError

    def param_block(self):
        """[system params | x_target | Q | R | Q_f] as float64 (include/ilqr_hip.h)."""
        n, m = self.n_x, self.n_u
        parts = [np.asarray(self._system_params(), dtype=np.float64).ravel(),
                 np.asarray(self.x_target, dtype=np.float64).reshape(n),
                 np.asarray(self.Q, dtype=np.float64).reshape(n * n),
                 np.asarray(self.R, dtype=np.float64).reshape(m * m),
                 np.asarray(self.Q_f, dtype=np.float64).reshape(n * n)]
        return np.concatenate(parts)

    def same_dynamics(self, other):
        """Same continuous dynamics and cost (the MPC plant may differ from the model in its integrator only)."""
        return np.array_equal(self.param_block(), other.param_block())

    def _set_cost(self, x_target, Q, R, Q_f):
        n, m = self.n_x, self.n_u
        self.x_target = np.asarray(x_target, dtype=np.float64).reshape(n)
        self.Q = np.asarray(Q, dtype=np.float64).reshape(n, n)
        self.R = np.asarray(R, dtype=np.float64).reshape(m, m)
        self.Q_f = np.asarray(Q_f, dtype=np.float64).reshape(n, n)

    def make_handle(self, horizon, batch, dtype=None, **kw):
        return _lib.Handle(system=self.SYSTEM_ID, n_x=self.n_x, n_u=self.n_u, horizon=horizon, batch=batch,
                           params=self.param_block(), dt=self.dt, integrator=self.integrator,
                           dtype=self.dtype if dtype is None else dtype, **kw)

    # ---- the 12 public callables (system_base.py:223-251), evaluated on the GPU ---------
    def _eval(self, name, x, u=None):
        if self._probe is None:
            self._probe = self.make_handle(horizon=1, batch=1)
        x = np.asarray(x, dtype=self.dtype)
        single = x.ndim == 1
        out = self._probe.eval_points(x, u, which=(name,))[name]
        return ready(out[0] if single else out)

    def f_fcn(self, x, u):
        return self._eval("f", x, u)

    def f_x_fcn(self, x, u):
        return self._eval("f_x", x, u)

    def f_u_fcn(self, x, u):
        return self._eval("f_u", x, u)

    def l_fcn(self, x, u):
        return self._eval("l", x, u)

    def l_x_fcn(self, x, u):
        return self._eval("l_x", x, u)

    def l_u_fcn(self, x, u):
        return self._eval("l_u", x, u)

    def l_xx_fcn(self, x, u):
        return self._eval("l_xx", x, u)

    def l_uu_fcn(self, x, u):
        return self._eval("l_uu", x, u)

    def l_ux_fcn(self, x, u):
        return self._eval("l_ux", x, u)

    def l_f_fcn(self, x):
        return self._eval("l_f", x)

    def l_f_x_fcn(self, x):
        return self._eval("l_f_x", x)

    def l_f_xx_fcn(self, x):
        return self._eval("l_f_xx", x)
