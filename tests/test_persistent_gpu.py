"""The persistent kernel (csrc/persistent.hpp: the whole iteration loop of a workgroup's trajectories in ONE launch -- head
of the solve, acceptance steps, linearise + sweep, all rollouts, and for MPC the plant step and the warm-start shift)
against the multi-launch forms (ILQR_FLAG_NO_PERSIST: one fused launch + one rollout launch per iteration under the
host's loop; ILQR_FLAG_NO_FUSE: linearise, sweep, rollout, select over the materialised expansion).

The phases are the same device functions, so everything is required to be IDENTICAL, bit for bit: iterate(n), solve
to convergence (a workgroup leaves its loop when ITS trajectories are done), repeated solves on warm state, and the
device-resident MPC loop (a workgroup's step lasts as long as its own slowest instance)."""
import numpy as np
import pytest

import ilqr_amd
from ilqr_amd import _lib, problems

pytestmark = pytest.mark.gpu

FIELDS = (("X", _lib.X), ("U", _lib.U), ("K", _lib.K), ("U_ff", _lib.UFF), ("cost", _lib.COST), ("alpha", _lib.ALPHA),
          ("status", _lib.STATUS), ("iters", _lib.ITERS), ("x0", _lib.X0))


def _handles(p, x0, U0, dtype, flags=0, **kw):
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], dtype)
    N = U0.shape[2]
    hs = [sysm.make_handle(horizon=N, batch=len(x0), n_alpha=10, n_trials=10, flags=flags | f, **kw)
          for f in (0, _lib.FLAG_NO_PERSIST, _lib.FLAG_NO_FUSE)]
    for h in hs:
        h.set_problem(x0, U0)
    return hs


def _same(hs, what):
    for name, f in FIELDS:
        ref = hs[0].get(f)
        for k, h in enumerate(hs[1:], 1):
            assert np.array_equal(ref, h.get(f), equal_nan=True), f"{what}: {name} differs between the persistent form and form {k}"


@pytest.mark.parametrize("B", [4, 37, 1040])          # 4-trajectory workgroups (one / several, partly filled) and 16-trajectory ones
def test_persistent_iterate_and_solve(B):
    p = problems.ua_double_pendulum(N=200)
    x0, U0 = problems.ua_batch(B, seed=5, restarts=True, N=200)
    hs = _handles(p, x0, U0, np.float32, tol=p["tol"], maxiter=50)
    for h in hs:
        h.initial_rollout()
        h.iterate(1)
    _same(hs, "one iteration")
    for h in hs:
        h.iterate(5)
    _same(hs, "five more in one call")
    its = [h.solve() for h in hs]
    for k in (1, 2):
        assert np.array_equal(its[0][0], its[k][0]) and np.array_equal(its[0][1], its[k][1])
    _same(hs, "solve")
    assert ((hs[0].get(_lib.STATUS) & 0xff) != _lib.TRAJ_ACTIVE).all()
    its = [h.solve() for h in hs]                   # warm state
    for k in (1, 2):
        assert np.array_equal(its[0][0], its[k][0]) and np.array_equal(its[0][1], its[k][1])
    _same(hs, "second solve")


def test_persistent_throughput_mode():
    p = problems.ua_double_pendulum(N=60)
    x0, U0 = problems.ua_batch(2048, seed=9, restarts=True, N=60)
    hs = _handles(p, x0, U0, np.float32, flags=_lib.FLAG_KEEP_ITERATING, maxiter=1 << 30)
    for h in hs:
        h.initial_rollout()
        h.iterate(6)
    _same(hs, "KEEP_ITERATING, 16-trajectory workgroups")


@pytest.mark.parametrize("integrator,N", [("backward_euler", 100), ("euler", 7), ("midpoint", 33)])
def test_persistent_pendulum(integrator, N):
    p = problems.pendulum_open_loop(integrator=integrator, N=N)
    rng = np.random.default_rng(3)
    for B in (20, 1100):        # (backward Euler has no 16-trajectory persistent form: B = 1100 runs the fused launches)
        x0 = np.tile(p["x0"], (B, 1)) + 0.1 * rng.standard_normal((B, 2))
        U0 = 0.1 * rng.standard_normal((B, 1, N))
        hs = _handles(p, x0, U0, np.float32, tol=p["tol"], maxiter=8)
        its = [h.solve() for h in hs]
        for k in (1, 2):
            assert np.array_equal(its[0][0], its[k][0]) and np.array_equal(its[0][1], its[k][1])
        _same(hs, f"{integrator} N={N} B={B}")


@pytest.mark.parametrize("B,maxiter", [(32, 6), (48, 20), (1040, 12)])
def test_persistent_mpc(B, maxiter):
    p = problems.ua_double_pendulum(N=50)
    x0, U0 = problems.ua_batch(B, seed=2, N=50)
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], np.float32)
    out = []
    for flags in (0, _lib.FLAG_NO_PERSIST, _lib.FLAG_NO_FUSE):
        h = sysm.make_handle(horizon=50, batch=B, n_alpha=10, n_trials=10, tol=p["tol"], maxiter=maxiter,
                             plant_integrator="backward_euler", flags=flags)
        h.mpc_reset(x0, U0)
        out.append((h.mpc_run(3), h.mpc_run(2), h))
    for k in (1, 2):
        for run in (0, 1):
            for q in range(3):
                assert np.array_equal(out[0][run][q], out[k][run][q]), (k, run, q)
    _same([o[2] for o in out], "state after the MPC steps")


def _dp_batch(B, N, seed=4):
    p = problems.double_pendulum(N=N)
    rng = np.random.default_rng(seed)
    x0 = np.asarray(p["x0"])[None] + 0.1 * rng.standard_normal((B, 4))
    U0 = 0.05 * rng.standard_normal((B, 2, N))
    return p, x0, U0


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("B", [37, 1040])
def test_fully_actuated_double_pendulum_all_forms(dtype, B):
    """n_x = 4, n_u = 2 (double_pendulum_sys.py): the 64-scalar tile of backward_tile16m2.hpp through the LDS ring -- fused
    and (fp32) persistent against the materialised kernels, every bit of the state; solve, warm solve, iterate."""
    p, x0, U0 = _dp_batch(B, 100)
    hs = _handles(p, x0, U0, dtype, tol=p["tol"], maxiter=30)
    for h in hs:
        h.initial_rollout()
        h.iterate(1)
    _same(hs, "one iteration")
    for h in hs:
        h.iterate(4)
    _same(hs, "four more in one call")
    its = [h.solve() for h in hs]
    for k in (1, 2):
        assert np.array_equal(its[0][0], its[k][0]) and np.array_equal(its[0][1], its[k][1])
    _same(hs, "solve")
    assert ((hs[0].get(_lib.STATUS) & 0xff) != _lib.TRAJ_ACTIVE).all()


@pytest.mark.parametrize("integrator,N", [("backward_euler", 30), ("euler", 9), ("midpoint", 64)])
def test_fully_actuated_other_integrators(integrator, N):
    p, x0, U0 = _dp_batch(24, N)
    p = problems.double_pendulum(integrator=integrator, N=N)
    for dtype in (np.float32, np.float64):
        hs = _handles(p, x0, U0, dtype, tol=p["tol"], maxiter=6)
        its = [h.solve() for h in hs]
        for k in (1, 2):
            assert np.array_equal(its[0][0], its[k][0]) and np.array_equal(its[0][1], its[k][1])
        _same(hs, f"(4,2) {integrator} N={N} {np.dtype(dtype).name}")


def test_fully_actuated_mpc():
    """run_MPC_double_pendulum.py's loop, device-resident: one persistent launch against the host-looped forms."""
    p, x0, U0 = _dp_batch(40, 40)
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], np.float32)
    out = []
    for flags in (0, _lib.FLAG_NO_PERSIST, _lib.FLAG_NO_FUSE):
        h = sysm.make_handle(horizon=40, batch=40, n_alpha=10, n_trials=10, tol=p["tol"], maxiter=10,
                             plant_integrator="backward_euler", flags=flags)
        h.mpc_reset(x0, U0)
        out.append((h.mpc_run(3), h))
    for k in (1, 2):
        for q in range(3):
            assert np.array_equal(out[0][0][q], out[k][0][q]), (k, q)
    _same([o[1] for o in out], "state after the MPC steps")
