"""Importable alias of the package directory ``iterative-linear-quadratic-regulator_amd``.

The directory name required by the build contract contains hyphens, which Python
cannot import directly; this stub makes ``import ilqr_amd`` (and
``ilqr_amd.systems.pendulum_sys`` ...) resolve to the files in that directory.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "iterative-linear-quadratic-regulator_amd")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _f
