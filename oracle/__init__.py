"""CPU oracle for the iLQR hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This package is a plain NumPy restatement (fp64 by default, fp32 switch) of the
reference algorithm in /root/reference/python/class_files/{iLQR_class.py,
systems/*.py}.  Every function cites the reference file:line it follows.

Who may import it: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- always as the checker / reported
baseline, never as the thing measured or shipped.  The product path
(``iterative-linear-quadratic-regulator_amd``) never imports this package and
fails loudly when the HIP library is missing.

PARITY PINNING STATUS
---------------------
The reference holds no tests, golden vectors or recorded outputs for this path
(SURVEY.md section 4 / 8c), and it cannot be executed here (it hard-imports
``jax`` which is not installed; ordinary ModuleNotFoundError).  So parity is
**unpinned by the reference's own fixtures**.  The oracle is instead pinned by
independent known answers (tests/test_oracle_*.py):

* closed-form Euler-discretised pendulum derivatives
  (matlab/CLASSES/Pendulum_System_CLASS.m:55-111),
* torch.func.jacfwd / hessian / grad (fp64, CPU) of an independent torch
  restatement of the dynamics and cost -- i.e. the same autodiff transforms the
  reference applies (system_base.py:203-219),
* central finite differences,
* the finite-horizon discrete Riccati recursion for a linear-quadratic problem
  (matlab/CLASSES/Linear_iLQR_CLASS.m:56-139) and one-iteration convergence,
* a second, independently written C restatement (oracle/c/ilqr_oracle.c).
"""

from .systems import (  # noqa: F401
    OracleSystem,
    PendulumOracle,
    UADoublePendulumOracle,
    DoublePendulumOracle,
    LinearQuadraticOracle,
    INTEGRATORS,
)
from .ilqr import (  # noqa: F401
    backward_pass,
    forward_pass,
    iLQROracle,
    mpc_closed_loop,
    horizon_steps,
)
