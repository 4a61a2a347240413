#!/usr/bin/env python3
"""A user-defined system end to end: what the reference asks of a `System` subclass
(python/class_files/systems/system_base.py:255-275 -- write _f_cont_fcn, and optionally _l_fcn / _l_f_fcn) on the
MI355X path.  The class below is everything the user writes; tracing, symbolic differentiation, code generation
and the hipcc build of the plugin happen on first use (cached in-tree by content hash).

    python scripts/run_iLQR_user_system.py [--batch B] [--dtype f64|f32] [--custom-cost]
"""
import argparse
import os
import sys
import time

import numpy as np
import sympy as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd.iLQR_class import iLQR                   # noqa: E402
from ilqr_amd.systems import SymbolicSystem            # noqa: E402


class Acrobot(SymbolicSystem):
    """Two-link arm actuated at the elbow only; x = [q1, q2, q1_dot, q2_dot] (q1 = 0 hanging), u = [elbow torque]."""

    def __init__(self, dt, custom_cost=False, **kw):
        self.m1 = self.m2 = 1.0
        self.l1 = self.l2 = 1.0
        self.g = 9.81
        self._custom = custom_cost
        target = [np.pi, 0.0, 0.0, 0.0]
        if custom_cost:
            super().__init__(4, 1, dt, **kw)
        else:
            super().__init__(4, 1, dt, x_target=target, Q=np.diag([1.0, 1.0, 0.1, 0.1]), R=[[0.1]],
                             Q_f=np.diag([500.0, 500.0, 50.0, 50.0]), **kw)

    def _f_cont_fcn(self, x, u):
        q1, q2, q1d, q2d = x
        m1, m2, l1, l2, g = self.m1, self.m2, self.l1, self.l2, self.g
        lc1, lc2, i1, i2 = l1 / 2, l2 / 2, m1 * l1 ** 2 / 12, m2 * l2 ** 2 / 12
        d11 = m1 * lc1 ** 2 + m2 * (l1 ** 2 + lc2 ** 2 + 2 * l1 * lc2 * sp.cos(q2)) + i1 + i2
        d12 = m2 * (lc2 ** 2 + l1 * lc2 * sp.cos(q2)) + i2
        d22 = m2 * lc2 ** 2 + i2
        h = m2 * l1 * lc2 * sp.sin(q2)
        phi2 = m2 * lc2 * g * sp.sin(q1 + q2)
        phi1 = (m1 * lc1 + m2 * l1) * g * sp.sin(q1) + phi2
        r1 = h * q2d ** 2 + 2 * h * q1d * q2d - phi1
        r2 = u[0] - h * q1d ** 2 - phi2
        det = d11 * d22 - d12 * d12
        return [q1d, q2d, (d22 * r1 - d12 * r2) / det, (d11 * r2 - d12 * r1) / det]


class AcrobotEnergyCost(Acrobot):
    """Same dynamics with a hand-written cost: distance of the tip from the upright position plus effort."""

    def _l_fcn(self, x, u):
        tip_height = -self.l1 * sp.cos(x[0]) - self.l2 * sp.cos(x[0] + x[1])
        return self.dt * (2.0 * (self.l1 + self.l2 - tip_height) + 0.05 * (x[2] ** 2 + x[3] ** 2) + 0.1 * u[0] ** 2)

    def _l_f_fcn(self, x):
        return 500.0 * ((x[0] - sp.pi) ** 2 + x[1] ** 2) + 50.0 * (x[2] ** 2 + x[3] ** 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--dtype", default="f32", choices=["f64", "f32"])
    ap.add_argument("--custom-cost", action="store_true")
    ap.add_argument("--maxiter", type=int, default=50)
    a = ap.parse_args()
    dtype = np.float64 if a.dtype == "f64" else np.float32
    dt, N = 0.02, 150
    t0 = time.time()
    system = (AcrobotEnergyCost if a.custom_cost else Acrobot)(dt, custom_cost=a.custom_cost, dtype=dtype, integrator="rk4")
    system.plugin_path(verbose=True)
    print(f"plugin ready in {time.time() - t0:.1f} s: {system.plugin_path()}")
    rng = np.random.default_rng(0)
    x_0 = np.array([0.0, 0.0, 0.0, 0.0])[None, :] + 0.05 * rng.standard_normal((a.batch, 4))
    U_init = 0.5 * rng.standard_normal((a.batch, 1, N))      # random restarts
    solver = iLQR(system=system, T=None, N=N, x_0=x_0, U_init=U_init, tol=1e-5, maxiter=a.maxiter, verbose=False)
    t0 = time.time()
    X, U, cost = solver.optimize_trajectory()
    el = time.time() - t0
    its = np.asarray(solver.iterations)
    print(f"{a.batch} restarts, N={N}: {el * 1e3:.1f} ms, iterations min/median/max {its.min()}/{int(np.median(its))}/{its.max()}")
    print(f"final cost min/median/max: {np.min(cost):.3f} / {np.median(cost):.3f} / {np.max(cost):.3f}")
    best = int(np.argmin(cost))
    print(f"best restart {best}: final state {np.asarray(X)[best, :, -1].round(3)}")


if __name__ == "__main__":
    main()
