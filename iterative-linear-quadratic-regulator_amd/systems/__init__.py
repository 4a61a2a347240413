"""Host-side mirror of the reference's ``class_files/systems`` package."""
from .system_base import System  # noqa: F401
from .pendulum_sys import MyPendulum  # noqa: F401
from .UA_double_pendulum_sys import MyUADoublePendulum  # noqa: F401
from .double_pendulum_sys import MyDoublePendulum  # noqa: F401
from .linear_sys import MyLinearSystem  # noqa: F401
from .custom_sys import SymbolicSystem  # noqa: F401
