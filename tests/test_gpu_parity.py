"""GPU parity tests: the HIP hot path (through the C-ABI) against the NumPy oracle on the
same seeded inputs.  Tolerance (north_star): rtol 1e-5 on K_t, k_t and total cost in the
fp64 parity mode; the fp32 (reference-precision) mode is checked at a looser, stated
tolerance because a 200-step fp32 Riccati recursion cannot reach 1e-5 with a different
operation order."""
import numpy as np
import pytest

import ilqr_amd
from ilqr_amd import _lib, problems
from oracle import backward_pass, forward_pass, iLQROracle, mpc_closed_loop
from oracle.build import oracle_from_spec, oracle_from_system

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def _specs():
    return {
        "pendulum": problems.pendulum_open_loop(N=100),
        "ua": problems.ua_double_pendulum(N=60),
        "dp": problems.double_pendulum(N=50),
    }


def _rand_traj(n, m, N, B, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((B, n, N + 1)) * scale, rng.standard_normal((B, m, N)) * scale


@pytest.mark.parametrize("name", ["pendulum", "ua", "dp"])
@pytest.mark.parametrize("integrator", ["euler", "midpoint", "rk4", "backward_euler"])
def test_system_callables_match_oracle(name, integrator):
    """The 12 System callables (system_base.py:223-251) at random points."""
    p = _specs()[name]
    dyn = dict(p["dynamics"], integrator=integrator)
    sysm = ilqr_amd.make_system(dyn, p["cost"])
    orc = oracle_from_system(sysm)
    rng = np.random.default_rng(7)
    n, m = sysm.n_x, sysm.n_u
    xs = rng.standard_normal((5, n)) * 1.5
    us = rng.standard_normal((5, m)) * 2.0
    for x, u in zip(xs, us):
        for fn in ("f_fcn", "f_x_fcn", "f_u_fcn", "l_fcn", "l_x_fcn", "l_u_fcn", "l_xx_fcn", "l_uu_fcn", "l_ux_fcn"):
            got, want = getattr(sysm, fn)(x, u), getattr(orc, fn)(x, u)
            np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-11, err_msg=f"{name} {integrator} {fn}")
        for fn in ("l_f_fcn", "l_f_x_fcn", "l_f_xx_fcn"):
            np.testing.assert_allclose(getattr(sysm, fn)(x), getattr(orc, fn)(x), rtol=1e-9, atol=1e-11)


def _close(got, want, rtol, what=""):
    """north_star tolerance as a matrix-level relative error: max|got - want| <= rtol * max|want|
    (entries of K_t that are ~0 carry no relative information of their own)."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-300)
    assert err <= rtol, f"{what}: relative error {err:.3e} > {rtol:g}"


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("name", ["pendulum", "ua", "dp"])
def test_backward_pass_matches_oracle(name, dtype):
    """K_t, k_t of one backward sweep around a random trajectory: rtol 1e-5, in the fp64 mode AND in the
    reference's own fp32 precision (inputs rounded to the mode's dtype before both sides see them)."""
    p = _specs()[name]
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], dtype)
    orc = oracle_from_system(sysm)
    N, B = p["N"], 5
    X, U = _rand_traj(sysm.n_x, sysm.n_u, N, B, seed=11, scale=0.7)
    X, U = X.astype(dtype).astype(np.float64), U.astype(dtype).astype(np.float64)
    s = ilqr_amd.iLQR(sysm, None, X[:, :, 0], U, N=N, verbose=False)
    uff, K = s.backward_pass(X, U)
    assert K.dtype == dtype
    for b in range(B):
        uff_o, K_o = backward_pass(orc, X[b], U[b])
        if dtype == np.float64:
            np.testing.assert_allclose(K[b], K_o, rtol=RTOL, atol=1e-9)
            np.testing.assert_allclose(uff[b], uff_o, rtol=RTOL, atol=1e-9)
        _close(K[b], K_o, RTOL, "K")
        _close(uff[b], uff_o, RTOL, "k")


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("name", ["pendulum", "ua", "dp"])
@pytest.mark.parametrize("alpha", [0.0, 1.0, 0.25])
def test_forward_pass_matches_oracle(name, alpha, dtype):
    p = _specs()[name]
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], dtype)
    orc = oracle_from_system(sysm)
    N, B = p["N"], 4
    n, m = sysm.n_x, sysm.n_u
    rng = np.random.default_rng(5)
    X, U = _rand_traj(n, m, N, B, seed=3, scale=0.3)
    uff = rng.standard_normal((B, m, N)) * 0.1
    K = rng.standard_normal((B, N, m, n)) * 0.1
    x0 = rng.standard_normal((B, n)) * 0.3
    rd = lambda a: a.astype(dtype).astype(np.float64)
    X, U, uff, K, x0 = rd(X), rd(U), rd(uff), rd(K), rd(x0)
    s = ilqr_amd.iLQR(sysm, None, x0, U, N=N, verbose=False)
    Xn, Un, c = s.forward_pass(x0, alpha, X, U, uff, K)
    for b in range(B):
        Xo, Uo, co = forward_pass(orc, x0[b], alpha, X[b], U[b], uff[b], K[b])
        np.testing.assert_allclose(c[b], co, rtol=RTOL)
        if dtype == np.float64:
            np.testing.assert_allclose(Xn[b], Xo, rtol=1e-6, atol=1e-8)
            np.testing.assert_allclose(Un[b], Uo, rtol=1e-6, atol=1e-8)
        else:
            _close(Xn[b], Xo, 1e-4, "X")


def test_linearize_tensor_matches_oracle():
    """ILQR_LIN (the materialised expansion the backward kernel streams) element by element."""
    p = _specs()["ua"]
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"])
    orc = oracle_from_system(sysm)
    N, B = 12, 3
    x0, U = problems.ua_batch(B, seed=2, restarts=True, N=N)
    s = ilqr_amd.iLQR(sysm, None, x0, U, N=N, verbose=False)
    h = s.handle
    h.initial_rollout()
    h.linearize()
    lin = h.get(_lib.LIN)
    X, Uc = h.get(_lib.X), h.get(_lib.U)
    for b in range(B):
        for t in range(N):
            x, u = X[b, :, t], Uc[b, :, t]
            want = np.concatenate([orc.f_x(x, u).ravel(), orc.f_u(x, u).ravel(), orc.l_x(x, u), orc.l_u(x, u),
                                   orc.l_xx(x, u).ravel(), orc.l_ux(x, u).ravel(), orc.l_uu(x, u).ravel()])
            np.testing.assert_allclose(lin[b, t], want, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("name,maxiter", [("pendulum", 20), ("ua", 8), ("dp", 8)])
def test_full_solve_matches_oracle(name, maxiter):
    """optimize_trajectory (iLQR_class.py:250-313) for a small batch: same accepted alphas,
    same iteration counts, same status, cost / K / k at rtol 1e-5."""
    p = _specs()[name]
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"])
    orc = oracle_from_system(sysm)
    N, B = p["N"], 3
    n, m = sysm.n_x, sysm.n_u
    rng = np.random.default_rng(21)
    x0 = np.asarray(p["x0"])[None, :] + rng.standard_normal((B, n)) * 0.1
    U0 = rng.standard_normal((B, m, N)) * 0.1
    s = ilqr_amd.iLQR(sysm, None, x0, U0, N=N, tol=p["tol"], maxiter=maxiter, verbose=False)
    X, U, cost = s.optimize_trajectory()
    K, uff = s.K, s.U_ff
    for b in range(B):
        o = iLQROracle(orc, N=N, x_0=x0[b], U_init=U0[b], tol=p["tol"], maxiter=maxiter)
        Xo, Uo, co = o.optimize_trajectory()
        assert s.status[b] == o.status
        assert int(s.iterations[b]) == o.iterations
        np.testing.assert_allclose(cost[b], co, rtol=RTOL)
        np.testing.assert_allclose(K[b], o.K, rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(uff[b], o.U_ff, rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(X[b], Xo, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(U[b], Uo, rtol=1e-5, atol=1e-7)


def test_unbatched_api_matches_reference_layouts():
    """B = 1 drop-in surface: shapes (n, N+1), (m, N), (N, m, n) and block_until_ready()."""
    p = problems.pendulum_open_loop(integrator="rk4", N=50)
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"])
    s = ilqr_amd.iLQR(sysm, 0.5, p["x0"], np.zeros((1, 50)), tol=1e-5, maxiter=5, verbose=False)
    assert s.N == 50
    Xw = np.zeros_like(s.X)
    Uw = np.zeros_like(s.U)
    uff, K = s.backward_pass(Xw, Uw)
    uff.block_until_ready()
    assert uff.shape == (1, 50) and K.shape == (50, 1, 2)
    Xn, Un, c = s.forward_pass(s.x_0, 0.0, Xw, Uw, uff * 0, K * 0)
    assert Xn.shape == (2, 51) and Un.shape == (1, 50) and np.ndim(c) == 0
    X, U, cost = s.optimize_trajectory()
    assert X.shape == (2, 51) and U.shape == (1, 50)
    o = iLQROracle(oracle_from_system(sysm), N=50, x_0=p["x0"], U_init=np.zeros((1, 50)), tol=1e-5, maxiter=5)
    _, _, co = o.optimize_trajectory()
    np.testing.assert_allclose(cost, co, rtol=RTOL)


def test_mpc_closed_loop_matches_oracle():
    """run_iLQR_MPC.py:116-143 incl. the state carried between solves."""
    p = problems.pendulum_mpc(N=40)
    n_sim = 6
    st = ilqr_amd.mpc_init(p["dynamics"], p["cost"], p["x0"], p["U_init"], plant_integrator=p["plant_integrator"],
                           N=40, tol=p["tol"], maxiter=p["maxiter"])
    U_sim, X_sim, costs = st.solver.mpc_run(n_sim)
    orc = oracle_from_spec(p["dynamics"], p["cost"])
    plant = oracle_from_spec(p["dynamics"], p["cost"], integrator=p["plant_integrator"])
    o = iLQROracle(orc, N=40, x_0=p["x0"], U_init=p["U_init"], tol=p["tol"], maxiter=p["maxiter"])
    Xo, Uo, co = mpc_closed_loop(o, plant, p["x0"], p["U_init"], n_sim)
    np.testing.assert_allclose(U_sim, Uo.T, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(X_sim, Xo[:, 1:].T, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(costs, co, rtol=RTOL)


def test_fp32_mode_meets_the_tolerance_at_full_horizon():
    """The reference's own precision (JAX default f32) at the north-star shape N = 200: K_t, k_t
    and total cost of the fp32 mode against the fp64 oracle, rtol 1e-5 (matrix-level), on seeded c3 inputs;
    and a full 8-iteration solve's cost."""
    from oracle.c_oracle import COracle
    p = problems.ua_double_pendulum(N=200)
    B = 32
    x0, U0 = problems.ua_batch(B, seed=1000)
    co = COracle(p["dynamics"], p["cost"])
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], np.float32)
    s = ilqr_amd.iLQR(sysm, None, x0, U0, N=200, maxiter=8, verbose=False)
    z = lambda *sh: np.zeros(sh)
    X, U, c = s.forward_pass(x0, 0.0, z(B, 4, 201), U0, z(B, 1, 200), z(B, 200, 1, 4))
    uff, K = s.backward_pass(X, U)
    assert uff.dtype == np.float32
    x0r = x0.astype(np.float32).astype(np.float64)
    for b in range(B):
        _, _, c_o = co.forward_pass(x0r[b], 0.0, z(4, 201), U0[b], z(1, 200), z(200, 1, 4))
        uff_o, K_o = co.backward_pass(np.asarray(X[b], np.float64), np.asarray(U[b], np.float64))
        _close(K[b], K_o, RTOL, "K")
        _close(uff[b], uff_o, RTOL, "k")
        _close(c[b], c_o, RTOL, "cost")
    _, _, cs = s.optimize_trajectory()
    for b in range(4):
        r = co.solve(x0r[b], U0[b], maxiter=8)
        _close(cs[b], r["cost"], RTOL, "solve cost")

# ---- config c5: synthetic linear-quadratic system on the wave-cooperative kernels (n_x > 4) --------------
@pytest.mark.parametrize("n,m,N", [(16, 8, 60), (8, 4, 33)])
def test_linear_quadratic_wave_kernels_match_oracle(n, m, N):
    p = problems.linear_quadratic(n=n, m=m, N=N)
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"])
    orc = oracle_from_system(sysm)
    B = 3
    x0, U0 = problems.lq_batch(B, n, m, N)
    rng = np.random.default_rng(4)
    X, U = rng.standard_normal((B, n, N + 1)), rng.standard_normal((B, m, N))
    s = ilqr_amd.iLQR(sysm, None, x0, U0, N=N, tol=1e-9, maxiter=4, verbose=False)
    uff, K = s.backward_pass(X, U)
    for b in range(B):
        uff_o, K_o = backward_pass(orc, X[b], U[b])
        np.testing.assert_allclose(K[b], K_o, rtol=RTOL, atol=1e-9)
        np.testing.assert_allclose(uff[b], uff_o, rtol=RTOL, atol=1e-9)
    Xs, Us, cost = s.optimize_trajectory()
    for b in range(B):
        o = iLQROracle(orc, N=N, x_0=x0[b], U_init=U0[b], tol=1e-9, maxiter=4)
        Xo, Uo, co = o.optimize_trajectory()
        np.testing.assert_allclose(cost[b], co, rtol=RTOL)
        np.testing.assert_allclose(Us[b], Uo, rtol=1e-5, atol=1e-8)
        assert o.history[0][1] == 1.0   # LQ: the full step is accepted and is optimal
    # known answer: iLQR gains of an LQ problem = finite-horizon discrete Riccati recursion
    A, Bm, dt = p["dynamics"]["A"], p["dynamics"]["B"], p["dynamics"]["dt"]
    Q, R, P = p["cost"]["Q"] * dt, p["cost"]["R"] * dt, p["cost"]["Q_f"].copy()
    Kd = s.K
    for t in range(N - 1, -1, -1):
        Kt = -np.linalg.solve(R + Bm.T @ P @ Bm, Bm.T @ P @ A)
        np.testing.assert_allclose(Kd[0, t], Kt, rtol=1e-6, atol=1e-10)
        P = Q + A.T @ P @ A + A.T @ P @ Bm @ Kt


@pytest.mark.parametrize("name", ["ua", "dp", "lq16"])
def test_levenberg_regularisation_matches_oracle(name):
    """mu > 0 (the build's extension): K = -(Q_uu + mu I)^-1 Q_ux with the full value update."""
    if name == "lq16":
        p = problems.linear_quadratic(n=16, m=8, N=25)
    else:
        p = _specs()[name]
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"])
    orc = oracle_from_system(sysm)
    N, B, mu = p["N"], 2, 0.37
    X, U = _rand_traj(sysm.n_x, sysm.n_u, N, B, seed=8, scale=0.5)
    s = ilqr_amd.iLQR(sysm, None, X[:, :, 0], U, N=N, verbose=False, mu=mu)
    uff, K = s.backward_pass(X, U)
    for b in range(B):
        uff_o, K_o = backward_pass(orc, X[b], U[b], mu=mu)
        np.testing.assert_allclose(K[b], K_o, rtol=RTOL, atol=1e-9)
        np.testing.assert_allclose(uff[b], uff_o, rtol=RTOL, atol=1e-9)


@pytest.mark.parametrize("B,N", [(1, 7), (5, 13), (17, 41), (67, 9)])
def test_ragged_batch_and_horizon_sizes(B, N):
    """Batch not a multiple of 4 / 16 / 64 and horizon not a multiple of the prefetch ring."""
    p = problems.ua_double_pendulum(N=N)
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"])
    orc = oracle_from_system(sysm)
    x0, U0 = problems.ua_batch(B, seed=B, restarts=True, N=N)
    s = ilqr_amd.iLQR(sysm, None, x0, U0, N=N, tol=1e-5, maxiter=3, verbose=False)
    X, U, cost = s.optimize_trajectory()
    for b in sorted({0, B // 2, B - 1}):
        o = iLQROracle(orc, N=N, x_0=x0[b], U_init=U0[b], tol=1e-5, maxiter=3)
        _, Uo, co = o.optimize_trajectory()
        np.testing.assert_allclose(cost[b], co, rtol=RTOL)
        np.testing.assert_allclose(s.K[b], o.K, rtol=1e-4, atol=1e-7)


def test_finished_trajectories_are_frozen():
    """A trajectory that converged (or failed its line search) keeps X, U, K, cost while the rest of
    the batch goes on iterating (the reference's `break`, iLQR_class.py:267-271, 304-307)."""
    p = problems.ua_double_pendulum(N=50)
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"])
    B = 8
    x0, U0 = problems.ua_batch(B, seed=5, restarts=True, N=50)
    x0 = x0 * np.linspace(0.0, 3.0, B)[:, None]          # from "already at rest" to strongly perturbed
    s = ilqr_amd.iLQR(sysm, None, x0, U0, N=50, tol=0.5, maxiter=25, verbose=False)
    s.optimize_trajectory()
    it = np.asarray(s.iterations)
    orc = oracle_from_system(sysm)
    its_o = []
    for b in range(B):
        o = iLQROracle(orc, N=50, x_0=x0[b], U_init=U0[b], tol=0.5, maxiter=25)
        _, _, co = o.optimize_trajectory()
        its_o.append(o.iterations)
        assert int(it[b]) == o.iterations and s.status[b] == o.status
        np.testing.assert_allclose(s.cost[b], co, rtol=RTOL)
    assert min(its_o) < max(its_o)      # the batch really did stop at different iterations


def test_golden_fixtures_on_device():
    """The committed golden vectors (tests/golden/*.npz, generator committed) through the C-ABI."""
    import glob
    import os
    gold = os.path.join(os.path.dirname(__file__), "golden")
    base = {"c1": problems.pendulum_open_loop(), "c2": problems.ua_double_pendulum(), "dp": problems.double_pendulum()}
    batch_cases = [os.path.join(gold, f + ".npz") for f in ("c1_pendulum_be", "c1_pendulum_rk4", "c2_ua_be", "c2_ua_rk4", "dp_rk4")]
    for path in batch_cases:
        g = np.load(path)
        b0 = base[os.path.basename(path).split("_")[0]]
        sysm = ilqr_amd.make_system(dict(b0["dynamics"], integrator=str(g["integrator"])), b0["cost"])
        N = int(g["N"])
        s = ilqr_amd.iLQR(sysm, None, g["x0"], g["U_init"], N=N, tol=float(g["tol"]), maxiter=int(g["maxiter"]),
                          verbose=False)
        uff, K = s.backward_pass(g["rollout_X"], g["rollout_U"])
        np.testing.assert_allclose(K, g["first_K"], rtol=RTOL, atol=1e-9)
        np.testing.assert_allclose(uff, g["first_Uff"], rtol=RTOL, atol=1e-9)
        X, U, cost = s.optimize_trajectory()
        np.testing.assert_allclose(cost, g["cost"], rtol=RTOL)
        assert list(s.status) == [str(x) for x in g["status"]]
        assert np.array_equal(np.asarray(s.iterations), g["iterations"])
    g = np.load(os.path.join(gold, "mpc_pendulum.npz"))
    p = problems.pendulum_mpc(N=int(g["N"]))
    st = ilqr_amd.mpc_init(p["dynamics"], p["cost"], p["x0"], p["U_init"], plant_integrator="midpoint",
                           N=int(g["N"]), tol=p["tol"], maxiter=p["maxiter"])
    U_sim, X_sim, costs = st.solver.mpc_run(int(g["n_sim"]))
    np.testing.assert_allclose(costs, g["cost"], rtol=RTOL)
    np.testing.assert_allclose(U_sim, g["U_sim"].T, rtol=1e-5, atol=1e-8)


def test_full_size_properties_c3():
    """BASELINE size (B = 4096, N = 200), where the NumPy oracle is too slow to run everything:
    size-independent properties -- (1) sampled trajectories agree with the C oracle; (2) costs never
    increase over iterations (acceptance rule); (3) the full-size backward sweep is bit-reproducible and
    matches the C oracle on sampled trajectories."""
    from oracle.c_oracle import COracle
    p = problems.ua_double_pendulum()
    B, N = 4096, 200
    x0, U0 = problems.ua_batch(B, seed=1000)
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"])
    s = ilqr_amd.iLQR(sysm, None, x0, U0, N=N, tol=1e-5, maxiter=6, verbose=False)
    h = s.handle
    h.initial_rollout()
    prev = h.get(_lib.COST)
    for _ in range(6):
        h.iterate(1)
        c = h.get(_lib.COST)
        assert np.all(c <= prev)
        prev = c
    co = COracle(p["dynamics"], p["cost"])
    for b in (0, 1234, 4095):
        r = co.solve(x0[b], U0[b], tol=1e-5, maxiter=6)
        np.testing.assert_allclose(prev[b], r["cost"], rtol=RTOL)
    # (3) backward sweep around the final trajectories at full size: bit-identical when repeated
    # (deterministic, no atomics), and equal to the C oracle on sampled trajectories
    X, U = h.get(_lib.X), h.get(_lib.U)
    uff_a, K_a = s.backward_pass(X, U)
    uff_b, K_b = s.backward_pass(X, U)
    assert np.array_equal(K_a, K_b) and np.array_equal(uff_a, uff_b)
    assert np.isfinite(K_a).all()
    for b in (0, 1234, 4095):
        uff_o, K_o = co.backward_pass(X[b], U[b])
        np.testing.assert_allclose(K_a[b], K_o, rtol=RTOL, atol=1e-9)
        np.testing.assert_allclose(uff_a[b], uff_o, rtol=RTOL, atol=1e-9)


def test_c4_mpc_instances_at_shard_size():
    """BASELINE config c4 at one GPU's shard: 1024 warm-started MPC instances of the under-actuated double
    pendulum (run_iLQR_UA_MPC.py:17-174: rk4 optimiser, backward_euler plant, maxiter 50 -> 6 here), a few
    receding-horizon steps on the device; sampled instances against the C oracle's closed loop."""
    from oracle.c_oracle import COracle
    p = problems.ua_double_pendulum(N=60)
    B, n_sim, maxiter = 1024, 3, 6
    x0, U0 = problems.ua_batch(B, seed=2, restarts=False, N=60)
    st = ilqr_amd.mpc_init(p["dynamics"], p["cost"], x0, U0, plant_integrator="backward_euler", N=60, tol=p["tol"],
                           maxiter=maxiter)
    U_sim, X_sim, costs = st.solver.mpc_run(n_sim)
    assert U_sim.shape == (n_sim, B, 1) and X_sim.shape == (n_sim, B, 4) and np.isfinite(costs).all()
    co = COracle(p["dynamics"], p["cost"])
    plant = COracle(p["dynamics"], p["cost"], integrator="backward_euler")
    for b in (0, 517, 1023):
        x, U_guess, state = x0[b].copy(), U0[b].copy(), None
        for k in range(n_sim):
            r = co.solve(x, U_guess, tol=p["tol"], maxiter=maxiter, state=state)
            u0 = r["U"][:, 0]
            x = plant.step(x, u0, jac=False)[0]
            np.testing.assert_allclose(U_sim[k, b], u0, rtol=1e-5, atol=1e-8)
            np.testing.assert_allclose(X_sim[k, b], x, rtol=1e-5, atol=1e-8)
            np.testing.assert_allclose(costs[k, b], r["cost"], rtol=RTOL)
            U_guess = np.concatenate([r["U"][:, 1:], r["U"][:, -1:]], axis=1)
            state = (r["X"], r["U_ff"], r["K"])


def _random_expansion(B, N, n, m, seed, dtype):
    """Time-varying, mildly contracting dynamics and positive-definite costs with cross terms."""
    rng = np.random.default_rng(seed)
    rd = lambda a: a.astype(dtype).astype(np.float64)
    f_x = rd(np.eye(n) * 0.95 + rng.standard_normal((B, N, n, n)) * (0.3 / np.sqrt(n)))
    f_u = rd(rng.standard_normal((B, N, n, m)) * 0.5)
    W = rng.standard_normal((B, N, n + m, n + m)) * 0.3
    H = W @ np.swapaxes(W, -1, -2) + np.eye(n + m) * 0.5
    l_xx, l_ux, l_uu = rd(H[..., :n, :n]), rd(H[..., n:, :n]), rd(H[..., n:, n:])
    l_x, l_u = rd(rng.standard_normal((B, N, n))), rd(rng.standard_normal((B, N, m)))
    Wf = rng.standard_normal((B, n, n))
    V_xx, V_x = rd(Wf @ np.swapaxes(Wf, -1, -2) + np.eye(n)), rd(rng.standard_normal((B, n)))
    return f_x, f_u, l_x, l_u, l_xx, l_ux, l_uu, V_x, V_xx


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,m,N", [(4, 1, 120), (2, 1, 50), (4, 2, 60), (3, 2, 40), (5, 1, 30), (6, 3, 30), (16, 8, 40),
                                   (11, 5, 25)])
def test_riccati_sweep_on_caller_supplied_tensors(n, m, N, dtype):
    """ilqr_backward_tensors: the sweep alone on an expansion the caller brings (generic / LQ mode), for the
    native kernel sizes and for sizes embedded by padding; K_t, k_t at rtol 1e-5 in both precisions."""
    from oracle.ilqr import backward_tensors
    B = 6
    ex = _random_expansion(B, N, n, m, seed=100 + n * 10 + m, dtype=dtype)
    sweep = ilqr_amd.RiccatiSweep(n, m, N, B, dtype=dtype)
    K, k = sweep(*ex)
    assert K.shape == (B, N, m, n) and k.shape == (B, m, N) and K.dtype == dtype
    for b in range(B):
        k_o, K_o = backward_tensors(*[a[b] for a in ex])
        _close(K[b], K_o, RTOL, f"K n={n} m={m}")
        _close(k[b], k_o, RTOL, f"k n={n} m={m}")


def test_tensor_sweep_reproduces_the_system_sweep():
    """Feeding the library's own linearisation (ILQR_LIN + the terminal derivatives) back through
    ilqr_backward_tensors gives bit-identical gains to ilqr_backward_pass: same kernel, same bytes."""
    p = _specs()["ua"]
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"])
    N, B = p["N"], 4
    X, U = _rand_traj(4, 1, N, B, seed=2, scale=0.5)
    s = ilqr_amd.iLQR(sysm, None, X[:, :, 0], U, N=N, verbose=False)
    uff, K = s.backward_pass(X, U)
    h = s.handle
    h.set(_lib.X, X)
    h.set(_lib.U, U)
    h.linearize()
    lin = h.get(_lib.LIN)
    term = np.stack([np.concatenate([sysm.l_f_x_fcn(X[b, :, -1]), sysm.l_f_xx_fcn(X[b, :, -1]).ravel()]) for b in range(B)])
    uff2, K2 = h.backward_tensors(lin, term)
    np.testing.assert_array_equal(K2, K)
    np.testing.assert_array_equal(uff2, uff)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("name", ["ua", "dp"])
def test_batch_members_do_not_see_each_other(name, dtype):
    """A trajectory solved inside a ragged batch (dead lanes in the last wave, members that stop early and freeze while
    the others go on) ends bit-identical to the same trajectory solved alone: nothing a dead or finished lane does --
    dropped out-of-range stores, predicated stores, the slot moves of the linearisation -- leaks into its neighbours."""
    N, B = 41, 67
    p = problems.ua_double_pendulum(N=N) if name == "ua" else problems.double_pendulum(N=N)
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], dtype)
    x0, U0 = problems.ua_batch(B, seed=3, restarts=True, N=N)
    if name == "dp":
        U0 = np.repeat(U0, 2, axis=1) * np.array([1.0, -0.5])[None, :, None]   # two controls
    x0 = x0 * np.linspace(0.0, 2.0, B)[:, None]
    s = ilqr_amd.iLQR(sysm, None, x0, U0, N=N, tol=0.3, maxiter=12, verbose=False)
    X, U, cost = s.optimize_trajectory()
    K, uff, it = s.K, s.U_ff, np.asarray(s.iterations)
    assert it.min() < it.max()
    for b in (0, 1, 31, 63, 64, 66):
        one = ilqr_amd.iLQR(sysm, None, x0[b:b + 1], U0[b:b + 1], N=N, tol=0.3, maxiter=12, verbose=False)
        X1, U1, c1 = one.optimize_trajectory()
        assert int(one.iterations[0]) == int(it[b]) and one.status[0] == s.status[b]
        np.testing.assert_array_equal(X[b], X1[0])
        np.testing.assert_array_equal(U[b], U1[0])
        np.testing.assert_array_equal(K[b], one.K[0])
        np.testing.assert_array_equal(uff[b], one.U_ff[0])
        np.testing.assert_array_equal(cost[b], c1[0])
