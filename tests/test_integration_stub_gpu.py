"""INTEGRATION.md section B shows the ctypes stub a maintainer of the reference would add next to iLQR_class.py.  This
test EXECUTES that text -- the code block is taken from the document, only the library path is made absolute -- on the
reference's open-loop pendulum problem (run_iLQR_open_loop.py:16-69) and checks it against the NumPy oracle, so the
document cannot drift from the C-ABI."""
import os
import re

import numpy as np
import pytest

import ilqr_amd
from ilqr_amd import _lib, problems
from oracle import backward_pass as o_backward, forward_pass as o_forward, iLQROracle
from oracle.build import oracle_from_spec

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_namespace():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## B."):]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    assert 'C.CDLL("libilqr_hip.so")' in code
    code = code.replace('C.CDLL("libilqr_hip.so")', f"C.CDLL({_lib.LIB_PATH!r})")
    ns = {}
    exec(compile(code, "INTEGRATION.md#B", "exec"), ns)
    return ns


def test_integration_md_stub_runs_and_matches_the_oracle():
    ns = _stub_namespace()
    N = 60
    p = problems.pendulum_open_loop(integrator="rk4", N=N)
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"])
    h = ns["make_handle"](_lib.SYS_PENDULUM, 2, 1, N, 1, sysm.param_block(), p["dynamics"]["dt"], _lib.INTEGRATORS["rk4"])
    orc = oracle_from_spec(p["dynamics"], p["cost"])
    rng = np.random.default_rng(0)
    U = 0.3 * rng.standard_normal((1, N))
    X, _, _ = o_forward(orc, p["x0"], 0.0, np.zeros((2, N + 1)), U, np.zeros((1, N)), np.zeros((N, 1, 2)))
    uff, K = ns["backward_pass"](h, X, U, 2, 1, N)
    uff_o, K_o = o_backward(orc, X, U)
    np.testing.assert_allclose(K, K_o, rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(uff, uff_o, rtol=1e-5, atol=1e-9)
    Xs, Us, cost = ns["optimize_trajectory"](h, p["x0"], p["U_init"][:, :N], 2, 1, N)
    o = iLQROracle(orc, N=N, x_0=p["x0"], U_init=p["U_init"][:, :N], tol=1e-5, maxiter=100)
    Xo, Uo, co = o.optimize_trajectory()
    np.testing.assert_allclose(cost, co, rtol=1e-5)
    np.testing.assert_allclose(Us, Uo, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(Xs, Xo, rtol=1e-4, atol=1e-6)
    assert ns["lib"].ilqr_destroy(h) == 0
