"""CPU-side checks of the user-defined-system path (reference contract
system_base.py:255-275): the oracle twins against symbolic differentiation, the code generator, the
plugin cross-compile and the argument checks of ilqr_create_custom.  No compute: that is tests/test_custom_gpu.py."""
import ctypes as C

import numpy as np
import pytest
import sympy as sp

from ilqr_amd import _lib
from ilqr_amd.systems import SymbolicSystem
from ilqr_amd.systems.custom_sys import generate_dyn_bodies
from ilqr_amd.systems.examples import example_problems
from oracle.custom import oracle_for_example

NAMES = ["sym_pendulum", "sym_ua", "cartpole", "unicycle", "dubins", "quadrotor", "swingup_cartpole", "obstacle_unicycle"]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_twin_matches_symbolic_statement(name):
    """The NumPy twin (independent statement + complex-step Jacobians) against the sympy statement the product
    compiles: pins both against a typo in either."""
    s, _, _ = example_problems()[name]
    o = oracle_for_example(name, s)
    xs, us = sp.symbols(f"x0:{s.n_x}"), sp.symbols(f"u0:{s.n_u}")
    f = sp.Matrix(s._f_cont_fcn(list(xs), list(us)))
    F, JX, JU = (sp.lambdify([xs, us], e, "numpy") for e in (f, f.jacobian(xs), f.jacobian(us)))
    rng = np.random.default_rng(0)
    for _ in range(6):
        x, u = rng.standard_normal(s.n_x) * 1.5, rng.standard_normal(s.n_u) * 2
        np.testing.assert_allclose(np.ravel(F(x, u)), o.f_cont(x, u), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(JX(x, u), o.f_cont_x(x, u), rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(JU(x, u), o.f_cont_u(x, u), rtol=1e-11, atol=1e-12)


@pytest.mark.parametrize("name", ["swingup_cartpole", "obstacle_unicycle"])
def test_oracle_cost_derivatives_match_symbolic_statement(name):
    """User costs: the oracle's complex-step / mixed finite-difference derivatives against sympy's exact ones."""
    s, _, _ = example_problems()[name]
    assert s.custom_cost
    o = oracle_for_example(name, s)
    n = s.n_x
    xs, us = sp.symbols(f"x0:{s.n_x}"), sp.symbols(f"u0:{s.n_u}")
    l, lf = s._l_fcn(list(xs), list(us)), s._l_f_fcn(list(xs))
    z = list(xs) + list(us)
    L, G, H = (sp.lambdify([xs, us], e, "numpy") for e in (l, sp.Matrix([l]).jacobian(z), sp.hessian(l, z)))
    LF, GF, HF = (sp.lambdify([xs], e, "numpy") for e in (lf, sp.Matrix([lf]).jacobian(xs), sp.hessian(lf, xs)))
    rng = np.random.default_rng(1)
    for _ in range(5):
        x, u = rng.standard_normal(s.n_x), rng.standard_normal(s.n_u)
        np.testing.assert_allclose(L(x, u), o.l(x, u), rtol=1e-13)
        np.testing.assert_allclose(np.ravel(G(x, u))[:n], o.l_x(x, u), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(np.ravel(G(x, u))[n:], o.l_u(x, u), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(H(x, u)[:n, :n], o.l_xx(x, u), rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(H(x, u)[n:, :n], o.l_ux(x, u), rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(H(x, u)[n:, n:], o.l_uu(x, u), rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(LF(x), o.l_f(x), rtol=1e-13)
        np.testing.assert_allclose(np.ravel(GF(x)), o.l_f_x(x), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(HF(x), o.l_f_xx(x), rtol=1e-7, atol=1e-8)


def test_generated_code_is_typed_and_complete():
    s, _, _ = example_problems()["cartpole"]
    f_body, fjac_body = generate_dyn_bodies(s._f_cont_fcn, 4, 1)
    for i in range(4):
        assert f"xd[{i}] =" in f_body and f"xd[{i}] =" in fjac_body
        assert f"Ju[{i}][0] =" in fjac_body
        for j in range(4):
            assert f"Jx[{i}][{j}] =" in fjac_body
    # every literal is wrapped in T(...): nothing promotes fp32 arithmetic to double
    import re
    bare = re.findall(r"(?<![\w.(\[])\d+\.\d+(?![\w.)])", fjac_body)
    assert not bare, bare
    assert "um::sin" in fjac_body and "std::" not in fjac_body


def test_bad_user_dynamics_are_rejected():
    class Wrong(SymbolicSystem):
        def _f_cont_fcn(self, x, u):
            return [x[1]]                       # n_x = 2 expected

    class Free(SymbolicSystem):
        def _f_cont_fcn(self, x, u):
            return [x[1], sp.Symbol("k") * u[0]]  # unbound constant

    class HalfCost(SymbolicSystem):
        def _f_cont_fcn(self, x, u):
            return [x[1], u[0]]

        def _l_fcn(self, x, u):                   # _l_f_fcn missing
            return x[0] ** 2 + u[0] ** 2

    for cls in (Wrong, Free):
        with pytest.raises(ValueError):
            cls(2, 1, 0.01, [0, 0], np.eye(2), [[1.0]], np.eye(2)).plugin_source()
    with pytest.raises(ValueError):
        HalfCost(2, 1, 0.01)
    with pytest.raises(ValueError):               # neither a quadratic cost nor user costs
        Wrong(2, 1, 0.01)
    with pytest.raises(ValueError):
        SymbolicSystem(7, 1, 0.01, np.zeros(7), np.eye(7), [[1.0]], np.eye(7))
    with pytest.raises(ValueError):  # same message as the reference's System (system_base.py:198)
        SymbolicSystem(2, 1, 0.01, [0, 0], np.eye(2), [[1.0]], np.eye(2), integrator="heun")


def test_plugin_cross_compiles_and_exports_its_entry_points():
    s, _, _ = example_problems(np.float32)["unicycle"]
    so = s.plugin_path()
    lib = C.CDLL(so)
    abi, nx, nu = C.c_int(), C.c_int(), C.c_int()
    assert lib.ilqr_plugin_info(C.byref(abi), C.byref(nx), C.byref(nu)) == 0
    assert (abi.value, nx.value, nu.value) == (_lib.ABI_VERSION, 3, 2)
    assert hasattr(lib, "ilqr_plugin_make_solver")
    assert s.plugin_path() == so  # cached by content hash


def test_create_custom_argument_checks():
    lib = _lib.load()
    assert lib.ilqr_param_count(_lib.SYS_CUSTOM, 4, 1) == 4 + 16 + 1 + 16   # the cost block only
    assert lib.ilqr_param_count(_lib.SYS_CUSTOM, 7, 1) == -1
    s, _, _ = example_problems()["cartpole"]
    with pytest.raises(ValueError, match="ilqr_create_custom"):     # a custom system without a plugin
        _lib.Handle(system=_lib.SYS_CUSTOM, n_x=4, n_u=1, horizon=5, batch=1, params=s.param_block(), dt=0.01,
                    integrator="rk4")
    with pytest.raises(ValueError, match="ilqr_create_custom"):     # a plugin with a built-in system id
        _lib.Handle(system=_lib.SYS_UA_DOUBLE_PENDULUM, n_x=4, n_u=1, horizon=5, batch=1,
                    params=np.zeros(9 + 4 + 16 + 1 + 16), dt=0.01, integrator="rk4", plugin="/nonexistent.so")
    h = C.c_void_p()
    assert lib.ilqr_create_custom(C.byref(h), None, b"/x.so") == _lib.ERR_INVALID_ARG
    cfg = _lib.Config()
    assert lib.ilqr_create_custom(C.byref(h), C.byref(cfg), None) == _lib.ERR_INVALID_ARG
