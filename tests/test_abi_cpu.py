"""CPU-side checks of the boundary: the C-ABI library builds, loads, exports every symbol
include/ilqr_hip.h declares, validates its arguments before touching a GPU, and -- with no GPU in
this container -- fails loudly instead of falling back to the CPU.  Host logic of the Python mirror
(shapes, ValueError behaviour of the reference) is covered here too."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import ilqr_amd
from ilqr_amd import _lib, problems

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    return _lib.device_count() == 0


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "ilqr_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(ilqr_\w+)\s*\(", header, flags=re.M))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ilqr_abi_version() == _lib.ABI_VERSION


def test_param_counts_and_support_matrix():
    lib = _lib.load()
    assert lib.ilqr_param_count(_lib.SYS_PENDULUM, 2, 1) == 3 + 2 + 4 + 1 + 4
    assert lib.ilqr_param_count(_lib.SYS_UA_DOUBLE_PENDULUM, 4, 1) == 9 + 4 + 16 + 1 + 16
    assert lib.ilqr_param_count(_lib.SYS_DOUBLE_PENDULUM, 4, 2) == 9 + 4 + 16 + 4 + 16
    assert lib.ilqr_param_count(_lib.SYS_LINEAR, 16, 8) == 256 + 128 + 16 + 256 + 64 + 256
    assert lib.ilqr_param_count(_lib.SYS_PENDULUM, 4, 1) == -1
    for dt in (_lib.F32, _lib.F64):
        assert lib.ilqr_is_supported(_lib.SYS_UA_DOUBLE_PENDULUM, 4, 1, dt) == 1
        assert lib.ilqr_is_supported(_lib.SYS_PENDULUM, 2, 1, dt) == 1
        assert lib.ilqr_is_supported(_lib.SYS_DOUBLE_PENDULUM, 4, 2, dt) == 1
    assert lib.ilqr_is_supported(_lib.SYS_PENDULUM, 2, 1, 7) == 0


def test_create_validates_arguments_before_touching_the_gpu():
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.ilqr_create(C.byref(h), None) == _lib.ERR_INVALID_ARG
    cfg = _lib.Config()
    cfg.struct_size = 3
    assert lib.ilqr_create(C.byref(h), C.byref(cfg)) == _lib.ERR_INVALID_ARG
    assert b"struct_size" in lib.ilqr_last_error(None)
    p = problems.pendulum_open_loop()
    s = ilqr_amd.make_system(p["dynamics"], p["cost"])
    with pytest.raises(ValueError):
        s.make_handle(horizon=0, batch=1)
    with pytest.raises(ValueError):
        s.make_handle(horizon=10, batch=1, n_alpha=17)
    with pytest.raises(ValueError, match="Unknown integrator"):
        _lib.Handle(system=_lib.SYS_PENDULUM, n_x=2, n_u=1, horizon=10, batch=1, params=s.param_block(), dt=0.01,
                    integrator="leapfrog")
    with pytest.raises(ValueError):   # wrong parameter block length
        _lib.Handle(system=_lib.SYS_PENDULUM, n_x=2, n_u=1, horizon=10, batch=1, params=np.zeros(5), dt=0.01,
                    integrator="rk4")


def test_no_cpu_fallback_without_a_gpu():
    if not _no_gpu():
        pytest.skip("a GPU is present")
    p = problems.pendulum_open_loop(N=20)
    s = ilqr_amd.make_system(p["dynamics"], p["cost"])
    with pytest.raises(_lib.IlqrError) as e:
        ilqr_amd.iLQR(s, None, p["x0"], np.zeros((1, 20)), N=20)
    assert e.value.code == _lib.ERR_NO_DEVICE and "no CPU fallback" in str(e.value)
    with pytest.raises(_lib.IlqrError):
        s.f_fcn(np.zeros(2), np.zeros(1))          # System callables also run on the device only


def test_reference_value_errors_come_from_host_logic():
    p = problems.pendulum_open_loop()
    with pytest.raises(ValueError, match="Unknown integrator: 'foo'"):       # system_base.py:198
        ilqr_amd.MyPendulum(0.01, [np.pi, 0], np.eye(2), np.eye(1), np.eye(2), integrator="foo")
    s = ilqr_amd.make_system(p["dynamics"], p["cost"])
    with pytest.raises(ValueError, match=r"U_init must have shape \(1, 100\), but got \(1, 50\)"):   # iLQR_class.py:50-52
        ilqr_amd.iLQR(s, 1.0, np.zeros(2), np.zeros((1, 50)))
    with pytest.raises(ValueError, match=r"U_init must have shape \(3, 1, 100\)"):
        ilqr_amd.iLQR(s, 1.0, np.zeros((3, 2)), np.zeros((1, 100)))
    assert ilqr_amd.horizon_steps(4.0, 0.01) == 400


def test_param_block_layout_matches_header():
    p = problems.ua_double_pendulum()
    s = ilqr_amd.make_system(p["dynamics"], p["cost"])
    blk = s.param_block()
    assert blk.shape == (_lib.load().ilqr_param_count(_lib.SYS_UA_DOUBLE_PENDULUM, 4, 1),)
    d = p["dynamics"]
    np.testing.assert_allclose(blk[:9], [d["g"], d["m1"], d["m2"], d["l1"], d["l2"], d["d1"], d["d2"], d["theta1"],
                                         d["theta2"]])
    np.testing.assert_allclose(blk[9:13], p["cost"]["x_target"])
    np.testing.assert_allclose(blk[13:29].reshape(4, 4), p["cost"]["Q"])
    np.testing.assert_allclose(blk[29], 50.0)
    np.testing.assert_allclose(blk[30:].reshape(4, 4), p["cost"]["Q_f"])
    lq = problems.linear_quadratic(n=4, m=2, N=10)
    s2 = ilqr_amd.make_system(lq["dynamics"], lq["cost"])
    assert s2.param_block().shape == (_lib.load().ilqr_param_count(_lib.SYS_LINEAR, 4, 2),)


def test_problem_generators_are_seeded_and_shaped():
    a, _ = problems.ua_batch(8, seed=0)
    b, U = problems.ua_batch(8, seed=0)
    assert np.array_equal(a, b) and a.shape == (8, 4) and U.shape == (8, 1, 200) and not U.any()
    lq = problems.linear_quadratic()
    A = lq["dynamics"]["A"]
    assert A.shape == (16, 16) and np.max(np.abs(np.linalg.eigvals(A))) <= 1.0 + 1e-12
