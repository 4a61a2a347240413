"""Generates the golden fixtures under tests/golden/ from the NumPy oracle (oracle/).

The reference cannot run here (it hard-imports jax, which is not installed: SURVEY.md 8c), and
it holds no fixtures of its own, so these vectors are produced by the oracle -- which is itself
pinned by independent known answers (tests/test_oracle_*.py).  They freeze the oracle's numbers
so that (a) any later edit of the oracle is caught, and (b) the GPU parity tests have inputs and
expected outputs that travel to the GPU box as plain data.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from ilqr_amd import problems  # noqa: E402
from oracle import backward_pass, forward_pass, iLQROracle, mpc_closed_loop  # noqa: E402
from oracle.build import oracle_from_spec  # noqa: E402


def case(name, p, N, B, maxiter, seed, restarts=True, integrator=None):
    dyn = dict(p["dynamics"])
    if integrator:
        dyn["integrator"] = integrator
    orc = oracle_from_spec(dyn, p["cost"])
    n, m = orc.n_x, orc.n_u
    rng = np.random.default_rng(seed)
    x0 = np.asarray(p["x0"], float)[None, :] + rng.standard_normal((B, n)) * 0.1
    U0 = rng.standard_normal((B, m, N)) * (0.1 if restarts else 0.0)
    out = dict(x0=x0, U_init=U0, N=N, maxiter=maxiter, tol=p["tol"], integrator=dyn["integrator"])
    X1, U1, c1, K1, k1, Xs, Us, cs, Ks, ks, its, sts, alphas = ([] for _ in range(13))
    for b in range(B):
        Xr, Ur, cr = forward_pass(orc, x0[b], 0.0, np.zeros((n, N + 1)), U0[b], np.zeros((m, N)), np.zeros((N, m, n)))
        uff, K = backward_pass(orc, Xr, Ur)
        X1.append(Xr); U1.append(Ur); c1.append(cr); K1.append(K); k1.append(uff)
        o = iLQROracle(orc, N=N, x_0=x0[b], U_init=U0[b], tol=p["tol"], maxiter=maxiter)
        X, U, c = o.optimize_trajectory()
        Xs.append(X); Us.append(U); cs.append(c); Ks.append(o.K); ks.append(o.U_ff)
        its.append(o.iterations); sts.append(o.status)
        alphas.append(np.array([h[1] for h in o.history] + [0.0] * (maxiter - len(o.history))))
    out.update(rollout_X=np.array(X1), rollout_U=np.array(U1), rollout_cost=np.array(c1), first_K=np.array(K1),
               first_Uff=np.array(k1), X=np.array(Xs), U=np.array(Us), cost=np.array(cs), K=np.array(Ks),
               U_ff=np.array(ks), iterations=np.array(its), status=np.array(sts), alphas=np.array(alphas))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "cost", np.array(cs), "iters", its, sts)


def mpc_warm_case():
    """run_iLQR_MPC.py as written: one full warm-up solve on the solver object (:95), then the loop (:116-143)."""
    p = problems.pendulum_mpc(N=40)
    orc = oracle_from_spec(p["dynamics"], p["cost"])
    plant = oracle_from_spec(p["dynamics"], p["cost"], integrator=p["plant_integrator"])
    o = iLQROracle(orc, N=40, x_0=p["x0"], U_init=p["U_init"], tol=p["tol"], maxiter=p["maxiter"])
    X, U, c = mpc_closed_loop(o, plant, p["x0"], p["U_init"], 8, warmup=True)
    np.savez_compressed(os.path.join(HERE, "mpc_pendulum_warm.npz"), X_sim=X, U_sim=U, cost=c, N=40, n_sim=8)
    print("mpc_pendulum_warm", c)


def open_loop_script_case():
    """c1 exactly as run_iLQR_open_loop.py runs it (:16-69): T = 4 s -> N = 400, backward_euler, one trajectory,
    x0 = [1, 0], U_init = 0, tol 1e-5, maxiter 100."""
    p = problems.pendulum_open_loop(integrator="backward_euler", N=400)
    orc = oracle_from_spec(p["dynamics"], p["cost"])
    o = iLQROracle(orc, N=400, x_0=p["x0"], U_init=p["U_init"], tol=p["tol"], maxiter=p["maxiter"])
    X, U, c = o.optimize_trajectory()
    np.savez_compressed(os.path.join(HERE, "c1_pendulum_be_n400.npz"), X=X, U=U, cost=c, K=o.K, U_ff=o.U_ff,
                        iterations=o.iterations, status=o.status, initial_cost=o.initial_cost,
                        alphas=np.array([h[1] for h in o.history]), costs=np.array([h[2] for h in o.history]))
    print("c1_pendulum_be_n400 cost", c, o.status, o.iterations)


def mpc_case():
    p = problems.pendulum_mpc(N=40)
    orc = oracle_from_spec(p["dynamics"], p["cost"])
    plant = oracle_from_spec(p["dynamics"], p["cost"], integrator=p["plant_integrator"])
    o = iLQROracle(orc, N=40, x_0=p["x0"], U_init=p["U_init"], tol=p["tol"], maxiter=p["maxiter"])
    X, U, c = mpc_closed_loop(o, plant, p["x0"], p["U_init"], 8)
    np.savez_compressed(os.path.join(HERE, "mpc_pendulum.npz"), X_sim=X, U_sim=U, cost=c, N=40, n_sim=8)
    print("mpc_pendulum", c)


if __name__ == "__main__":
    case("c1_pendulum_be", problems.pendulum_open_loop(), N=100, B=2, maxiter=15, seed=1, restarts=False)
    case("c1_pendulum_rk4", problems.pendulum_open_loop(integrator="rk4"), N=100, B=2, maxiter=15, seed=2)
    case("c2_ua_rk4", problems.ua_double_pendulum(), N=60, B=3, maxiter=8, seed=3)
    case("c2_ua_be", problems.ua_double_pendulum(integrator="backward_euler"), N=40, B=2, maxiter=6, seed=4)
    case("dp_rk4", problems.double_pendulum(), N=40, B=2, maxiter=6, seed=5)
    mpc_case()
    mpc_warm_case()
    open_loop_script_case()
