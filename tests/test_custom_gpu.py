"""GPU parity of user-defined systems (reference contract system_base.py:255-275): dynamics
written as a ``SymbolicSystem`` subclass, compiled into a plugin, run through ilqr_create_custom -- against oracle
twins that share none of that machinery (oracle/custom.py).  Same tolerances as tests/test_gpu_parity.py."""
import numpy as np
import pytest

import ilqr_amd
from ilqr_amd import problems
from ilqr_amd.systems.examples import example_problems, SymbolicPendulum
from oracle import backward_pass, forward_pass, iLQROracle, mpc_closed_loop
from oracle.custom import oracle_for_example

pytestmark = pytest.mark.gpu
NAMES = ["sym_pendulum", "sym_ua", "cartpole", "unicycle", "dubins", "quadrotor", "swingup_cartpole", "obstacle_unicycle"]
RTOL = 1e-5


def _close(got, want, rtol, what=""):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-300)
    assert err <= rtol, f"{what}: relative error {err:.3e} > {rtol:g}"


@pytest.mark.parametrize("integrator", ["euler", "midpoint", "rk4", "backward_euler"])
@pytest.mark.parametrize("name", NAMES)
def test_custom_system_callables(name, integrator):
    """f, f_x, f_u (chain rule through every integrator of the generated Jacobians) and the costs."""
    s, _, _ = example_problems(integrator=integrator)[name]
    o = oracle_for_example(name, s)
    rng = np.random.default_rng(3)
    for _ in range(4):
        x, u = rng.standard_normal(s.n_x) * 0.8, rng.standard_normal(s.n_u) * 1.5
        for fn in ("f_fcn", "f_x_fcn", "f_u_fcn", "l_fcn", "l_x_fcn", "l_u_fcn", "l_xx_fcn", "l_uu_fcn", "l_ux_fcn"):
            # (the oracle's second derivatives of a user cost carry its finite-difference error, ~1e-9)
            np.testing.assert_allclose(getattr(s, fn)(x, u), getattr(o, fn)(x, u), rtol=1e-7, atol=1e-9,
                                       err_msg=f"{name} {integrator} {fn}")
        for fn in ("l_f_fcn", "l_f_x_fcn", "l_f_xx_fcn"):
            np.testing.assert_allclose(getattr(s, fn)(x), getattr(o, fn)(x), rtol=1e-7, atol=1e-8)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("name", NAMES)
def test_custom_backward_and_forward_pass(name, dtype):
    s, N, _ = example_problems(dtype)[name]
    o = oracle_for_example(name, s)
    n, m, B = s.n_x, s.n_u, 5
    rng = np.random.default_rng(17)
    rd = lambda a: a.astype(dtype).astype(np.float64)
    X, U = rd(rng.standard_normal((B, n, N + 1)) * 0.5), rd(rng.standard_normal((B, m, N)) * 0.5)
    sol = ilqr_amd.iLQR(s, None, X[:, :, 0], U, N=N, verbose=False)
    uff, K = sol.backward_pass(X, U)
    for b in range(B):
        uff_o, K_o = backward_pass(o, X[b], U[b])
        _close(K[b], K_o, RTOL, f"{name} K")
        _close(uff[b], uff_o, RTOL, f"{name} k")
    uf, Kf = rd(rng.standard_normal((B, m, N)) * 0.05), rd(rng.standard_normal((B, N, m, n)) * 0.05)
    Xs = X * 0.3
    for alpha in (1.0, 0.25):
        Xn, Un, c = sol.forward_pass(Xs[:, :, 0], alpha, Xs, U, uf, Kf)
        for b in range(B):
            Xo, Uo, co = forward_pass(o, Xs[b, :, 0], alpha, Xs[b], U[b], uf[b], Kf[b])
            np.testing.assert_allclose(c[b], co, rtol=RTOL)
            _close(Xn[b], Xo, 1e-6 if dtype == np.float64 else 1e-4, f"{name} X")


@pytest.mark.parametrize("name,maxiter", [("sym_pendulum", 15), ("cartpole", 10), ("unicycle", 12), ("dubins", 10), ("quadrotor", 8),
                                          ("swingup_cartpole", 10), ("obstacle_unicycle", 10)])
def test_custom_full_solve(name, maxiter):
    """optimize_trajectory on a user system: same accepted alphas / iteration counts / status as the oracle."""
    s, N, x0 = example_problems()[name]
    o = oracle_for_example(name, s)
    B = 3
    rng = np.random.default_rng(5)
    x0s = x0[None, :] + rng.standard_normal((B, s.n_x)) * 0.05
    U0 = rng.standard_normal((B, s.n_u, N)) * 0.1 + (0.5 * 0.5 * 9.81 if name == "quadrotor" else 0.0)
    sol = ilqr_amd.iLQR(s, None, x0s, U0, N=N, tol=1e-4, maxiter=maxiter, verbose=False)
    X, U, cost = sol.optimize_trajectory()
    for b in range(B):
        ref = iLQROracle(o, N=N, x_0=x0s[b], U_init=U0[b], tol=1e-4, maxiter=maxiter)
        Xo, Uo, co = ref.optimize_trajectory()
        assert sol.status[b] == ref.status and int(sol.iterations[b]) == ref.iterations
        np.testing.assert_allclose(cost[b], co, rtol=RTOL)
        np.testing.assert_allclose(sol.K[b], ref.K, rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(X[b], Xo, rtol=1e-5, atol=1e-6)


def test_symbolic_pendulum_reproduces_the_builtin_system():
    """The reference's own pendulum written as a user system takes the same iterations to the same answer as the
    built-in kernel (they differ only in the sine implementation)."""
    p = problems.pendulum_open_loop(N=100)
    builtin = ilqr_amd.make_system(p["dynamics"], p["cost"])
    d, c = p["dynamics"], p["cost"]
    user = SymbolicPendulum(d["dt"], c["x_target"], c["Q"], c["R"], c["Q_f"], g=d.get("g", 9.81), l=d.get("l", 1.0),
                            d=d.get("d", 0.01), integrator=d.get("integrator", "rk4"))
    x0 = np.asarray(p["x0"])[None, :]
    U0 = np.zeros((1, 1, 100))
    out = []
    for sysm in (builtin, user):
        sol = ilqr_amd.iLQR(sysm, None, x0, U0, N=100, tol=p["tol"], maxiter=40, verbose=False)
        X, U, cost = sol.optimize_trajectory()
        out.append((X, U, cost, int(sol.iterations[0]), sol.status[0]))
    assert out[0][3] == out[1][3] and out[0][4] == out[1][4]
    np.testing.assert_allclose(out[1][2], out[0][2], rtol=1e-9)
    np.testing.assert_allclose(out[1][0], out[0][0], rtol=1e-7, atol=1e-9)


def test_custom_mpc_closed_loop():
    """MPC (run_iLQR_MPC.py:80-118) with a user system as both the model and, at a finer integrator, the plant."""
    s, _, x0 = example_problems(integrator="euler")["cartpole"]
    plant, _, _ = example_problems(integrator="rk4")["cartpole"]
    o, op = oracle_for_example("cartpole", s), oracle_for_example("cartpole", plant)
    N, n_sim = 30, 12
    U0 = np.zeros((1, N))
    sol = ilqr_amd.iLQR(s, None, x0, U0, N=N, tol=1e-3, maxiter=5, verbose=False, plant=plant)
    sol.mpc_reset(x0, U0)
    U_sim, X_sim, costs = sol.mpc_run(n_sim)
    ref = iLQROracle(o, N=N, x_0=x0, U_init=U0, tol=1e-3, maxiter=5)
    Xo, Uo, co = mpc_closed_loop(ref, op, x0, U0, n_sim)
    np.testing.assert_allclose(U_sim, Uo.T, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(X_sim, Xo[:, 1:].T, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(costs, co, rtol=RTOL)
    other, _, _ = example_problems(integrator="rk4")["cartpole"]
    other.length = 0.7  # a plant with different physics is refused, as for the built-in systems
    with pytest.raises(ValueError):
        ilqr_amd.iLQR(s, None, x0, U0, N=N, verbose=False, plant=other)


def test_custom_system_at_batch_scale():
    """4096 cart-pole restarts in fp32 on the DPP backward sweep: the batch's gains equal a per-trajectory fp64
    oracle on sampled trajectories, and the solve lowers every trajectory's cost."""
    s, N, x0 = example_problems(np.float32)["cartpole"]
    o = oracle_for_example("cartpole", s)
    B = 4096
    rng = np.random.default_rng(9)
    x0s = (x0[None, :] + rng.standard_normal((B, 4)) * 0.1).astype(np.float32)
    U0 = (rng.standard_normal((B, 1, N)) * 0.1).astype(np.float32)
    sol = ilqr_amd.iLQR(s, None, x0s, U0, N=N, tol=1e-4, maxiter=3, verbose=False)
    z = np.zeros
    X0, _, c0 = sol.forward_pass(x0s, 0.0, z((B, 4, N + 1), np.float32), U0, z((B, 1, N), np.float32),
                                 z((B, N, 1, 4), np.float32))   # alpha = 0: the open-loop rollout of U0
    X0, c0 = np.asarray(X0), np.asarray(c0)
    uff, K = sol.backward_pass(X0, U0)
    for b in (0, 17, 2048, 4095):
        uff_o, K_o = backward_pass(o, X0[b].astype(np.float64), U0[b].astype(np.float64))
        _close(K[b], K_o, RTOL, "K")
        _close(uff[b], uff_o, RTOL, "k")
    _, _, cost = sol.optimize_trajectory()
    assert np.isfinite(cost).all()
    assert (np.asarray(cost) <= c0 * (1 + 1e-6)).all()
