"""MI355X-native batched iLQR: the hot path of
MohamedAbou-Taleb/Iterative-Linear-Quadratic-Regulator (per-timestep linearisation,
backward Riccati sweep, line-searched forward rollout) as hand-written HIP kernels
for gfx950 behind a C-ABI, with a host-side mirror of the reference's Python
interface (``System`` subclasses, ``iLQR``) so its driver scripts run on top of it.

Import name: ``ilqr_amd`` (this directory's name is not a valid Python identifier).
"""

from . import _lib  # noqa: F401
from .iLQR_class import iLQR, horizon_steps  # noqa: F401
from .systems import (System, MyPendulum, MyUADoublePendulum, MyDoublePendulum,  # noqa: F401
                      MyLinearSystem)
from .api import solve, SolveResult, MPCState, mpc_init, mpc_step, make_system, RiccatiSweep  # noqa: F401
from . import problems  # noqa: F401
from . import dist  # noqa: F401

__all__ = ["iLQR", "horizon_steps", "System", "MyPendulum", "MyUADoublePendulum", "MyDoublePendulum",
           "MyLinearSystem", "solve", "SolveResult", "MPCState", "mpc_init", "mpc_step", "make_system", "RiccatiSweep",
           "problems"]
