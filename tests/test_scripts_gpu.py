"""The two drop-in drivers (north_star: "drops in under run_iLQR_open_loop.py and run_iLQR_MPC.py") run under test:
scripts/run_iLQR_open_loop.py with its default arguments, and scripts/run_iLQR_MPC.py both as the reference's
host loop through attribute writes (solver.x_0 = ..., solver.U = ..., optimize_trajectory(), plant f_fcn, shift --
run_iLQR_MPC.py:116-143) and as the device-resident loop, each against the NumPy oracle's closed loop INCLUDING the
driver's stateful warm-up solve (run_iLQR_MPC.py:95)."""
import importlib.util
import os

import numpy as np
import pytest

from ilqr_amd import problems
from oracle import iLQROracle, mpc_closed_loop
from oracle.build import oracle_from_spec

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _script(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "scripts", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_open_loop_driver_default_arguments(tmp_path, capsys):
    """`python scripts/run_iLQR_open_loop.py` as the reference runs it (N = 400, backward_euler, batch 1, verbose),
    with the figure: final cost / X / U / K against the committed golden vector; the reference's printed lines."""
    fig = tmp_path / "open_loop.png"
    out = _script("run_iLQR_open_loop").main(["--plot", str(fig)])
    g = np.load(os.path.join(GOLD, "c1_pendulum_be_n400.npz"))
    assert out["N"] == 400 and out["status"] == str(g["status"]) and out["iterations"] == int(g["iterations"])
    np.testing.assert_allclose(out["cost"], g["cost"], rtol=1e-5)
    np.testing.assert_allclose(out["X"], g["X"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(out["U"], g["U"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(out["K"], g["K"], rtol=1e-4, atol=1e-7)
    assert fig.exists() and fig.stat().st_size > 10000
    text = capsys.readouterr().out
    assert f"Initial cost: {float(g['initial_cost']):.4f}" in text       # iLQR_class.py:262
    for (a, c) in zip(g["alphas"], g["costs"]):                            # :296
        assert f"(alpha={a:.2e}): Cost improved to {c:.4f}" in text
    assert "Converged at iteration" in text and "Time taken to execute iLQR" in text


def _oracle_loop(N_h, n_sim, warmup):
    p = problems.pendulum_mpc(N=N_h)
    orc = oracle_from_spec(p["dynamics"], p["cost"])
    plant = oracle_from_spec(p["dynamics"], p["cost"], integrator=p["plant_integrator"])
    o = iLQROracle(orc, N=N_h, x_0=p["x0"], U_init=p["U_init"], tol=p["tol"], maxiter=p["maxiter"])
    return mpc_closed_loop(o, plant, p["x0"], p["U_init"], n_sim, warmup=warmup)


@pytest.mark.parametrize("mode", ["host-loop", "device"])
def test_mpc_driver_matches_the_oracle_with_its_warmup(mode, tmp_path):
    """`python scripts/run_iLQR_MPC.py [--host-loop] --steps 6`: the reference's horizon (T = 2 s, N = 200, optimiser
    backward_euler, plant midpoint, maxiter 10) with its warm-up solve carried into step 0."""
    args = ["--steps", "6", "--plot", str(tmp_path / "mpc.png")] + (["--host-loop"] if mode == "host-loop" else [])
    out = _script("run_iLQR_MPC").main(args)
    assert out["N_h"] == 200 and out["X_sim"].shape == (2, 7) and out["U_sim"].shape == (1, 6)
    Xo, Uo, co = _oracle_loop(200, 6, warmup=True)
    np.testing.assert_allclose(out["U_sim"], Uo, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(out["X_sim"], Xo, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(out["cost"], co, rtol=1e-5)
    assert (tmp_path / "mpc.png").stat().st_size > 10000
    # the warm-up is part of the result: the cold-start loop gives a different first control
    _, U_cold, _ = _oracle_loop(200, 1, warmup=False)
    assert abs(U_cold[0, 0] - Uo[0, 0]) > 1e-6 * max(1.0, abs(Uo[0, 0]))


@pytest.mark.parametrize("mode", ["host-loop", "device"])
def test_mpc_driver_against_the_warm_golden(mode):
    """The committed golden vector of the warm-started loop (tests/golden/mpc_pendulum_warm.npz: N = 40, 8 steps)."""
    g = np.load(os.path.join(GOLD, "mpc_pendulum_warm.npz"))
    args = ["--horizon", "0.4", "--steps", str(int(g["n_sim"]))] + (["--host-loop"] if mode == "host-loop" else [])
    out = _script("run_iLQR_MPC").main(args)
    assert out["N_h"] == int(g["N"])
    np.testing.assert_allclose(out["U_sim"], g["U_sim"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(out["X_sim"], g["X_sim"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(out["cost"], g["cost"], rtol=1e-5)


def test_ua_mpc_driver_cold_start_batch():
    """`--system ua` (run_iLQR_UA_MPC.py: rk4 optimiser, backward_euler plant, maxiter 50, side-effect-free warm-up)
    for a small batch on the device: instance 0 against the C oracle's closed loop."""
    from oracle.c_oracle import COracle
    out = _script("run_iLQR_MPC").main(["--system", "ua", "--batch", "5", "--steps", "3"])
    p = problems.ua_double_pendulum(N=200)
    rng = np.random.default_rng(2)
    x0 = np.zeros(4)[None, :] + rng.standard_normal((5, 4)) * 0.05
    co = COracle(p["dynamics"], p["cost"])
    plant = COracle(p["dynamics"], p["cost"], integrator="backward_euler")
    x, U_guess, state = x0[0].copy(), np.zeros((1, 200)), None
    for k in range(3):
        r = co.solve(x, U_guess, tol=p["tol"], maxiter=p["maxiter"], state=state)
        x = plant.step(x, r["U"][:, 0], jac=False)[0]
        np.testing.assert_allclose(out["U_sim"][k, 0], r["U"][:, 0], rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(out["X_sim"][k, 0], x, rtol=1e-5, atol=1e-8)
        U_guess = np.concatenate([r["U"][:, 1:], r["U"][:, -1:]], axis=1)
        state = (r["X"], r["U_ff"], r["K"])
