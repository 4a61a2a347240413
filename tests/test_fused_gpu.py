"""The fused iteration (csrc/backward_fused16.hpp: acceptance step + linearisation + sweep as one kernel, tiles handed
from producer waves to the sweep waves through LDS) against the materialised path (linearize_kernel ->
backward_tile16_kernel -> select_kernel over the expansion in HBM; ILQR_FLAG_NO_FUSE).

Both run the same device functions (Stepper::step_jac, tile16_pack, the DPP step, select_candidates) on the same
numbers, so the results are required to be IDENTICAL -- every bit of X, U, K, U_ff, cost, accepted alpha, status and
iteration count, after every iteration -- not close.  The parity of either path with the oracle is the business of
test_gpu_parity.py / test_gpu_fullshape.py, which run the fused path by default.
"""
import numpy as np
import pytest

import ilqr_amd
from ilqr_amd import _lib, problems

pytestmark = pytest.mark.gpu

FIELDS = (("X", _lib.X), ("U", _lib.U), ("K", _lib.K), ("U_ff", _lib.UFF), ("cost", _lib.COST), ("alpha", _lib.ALPHA),
          ("status", _lib.STATUS), ("iters", _lib.ITERS))


def _pair(p, x0, U0, dtype, **kw):
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], dtype)
    N = U0.shape[2]
    flags = kw.pop("flags", 0)
    hf = sysm.make_handle(horizon=N, batch=len(x0), n_alpha=10, n_trials=10, flags=flags, **kw)
    hm = sysm.make_handle(horizon=N, batch=len(x0), n_alpha=10, n_trials=10, flags=flags | _lib.FLAG_NO_FUSE, **kw)
    for h in (hf, hm):
        h.set_problem(x0, U0)
    return hf, hm


def _same(hf, hm, what):
    for name, f in FIELDS:
        a, b = hf.get(f), hm.get(f)
        assert np.array_equal(a, b, equal_nan=True), f"{what}: {name} differs (max |d| = {np.nanmax(np.abs(a - b.astype(a.dtype))):.3e})"


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("B", [64, 37])
def test_fused_iterations_are_bit_identical(dtype, B):
    """UA double pendulum (c3's system), rk4, N = 200; B = 37 leaves a partly filled workgroup and an idle sweep wave.
    Stepped one iteration at a time (the host reads in between, so every iteration's acceptance step is flushed by the
    stand-alone kernel) and then several at once (the fused launch runs it)."""
    p = problems.ua_double_pendulum(N=200)
    x0, U0 = problems.ua_batch(B, seed=5, restarts=True, N=200)
    hf, hm = _pair(p, x0, U0, dtype, tol=p["tol"], maxiter=50)
    for h in (hf, hm):
        h.initial_rollout()
    for it in range(3):
        hf.iterate(1)
        hm.iterate(1)
        _same(hf, hm, f"iteration {it}")
    hf.iterate(7)
    hm.iterate(7)
    _same(hf, hm, "after 7 more iterations in one call")
    tf, tm = hf.timing_get(), hm.timing_get()   # (timing off: all zero) -- the call must work with the new phase
    assert set(tf) == set(_lib.PHASES) and set(tm) == set(_lib.PHASES)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fused_solve_matches_materialised_solve(dtype):
    """ilqr_solve to convergence (tol 1e-5, maxiter 50): trajectories leave the loop at different iterations, so the
    fused kernel runs with partly finished workgroups, and the host stops two iterations after the last one left."""
    p = problems.ua_double_pendulum(N=200)
    x0, U0 = problems.ua_batch(96, seed=0, N=200)
    hf, hm = _pair(p, x0, U0, dtype, tol=p["tol"], maxiter=50)
    itf, cf = hf.solve()
    itm, cm = hm.solve()
    assert np.array_equal(itf, itm) and np.array_equal(cf, cm)
    _same(hf, hm, "solve")
    assert ((hf.get(_lib.STATUS) & 0xff) != _lib.TRAJ_ACTIVE).all()
    # and a second solve on the same handles (warm state, the head of the solve ignores the previous statuses)
    itf, cf = hf.solve()
    itm, cm = hm.solve()
    assert np.array_equal(itf, itm) and np.array_equal(cf, cm)
    _same(hf, hm, "second solve")


@pytest.mark.parametrize("integrator,N", [("backward_euler", 400), ("rk4", 101), ("euler", 7), ("midpoint", 3)])
def test_fused_pendulum_other_integrators_and_horizons(integrator, N):
    """n_x = 2 on the zero-padded tile, the other integrators, horizons that are not a multiple of the producers' unit
    (4 time steps) or shorter than the LDS ring."""
    p = problems.pendulum_open_loop(integrator=integrator, N=N)
    rng = np.random.default_rng(3)
    B = 20
    x0 = np.tile(p["x0"], (B, 1)) + 0.1 * rng.standard_normal((B, 2))
    U0 = 0.1 * rng.standard_normal((B, 1, N))
    for dtype in (np.float32, np.float64):
        hf, hm = _pair(p, x0, U0, dtype, tol=p["tol"], maxiter=8)
        itf, cf = hf.solve()
        itm, cm = hm.solve()
        assert np.array_equal(itf, itm) and np.array_equal(cf, cm)
        _same(hf, hm, f"{integrator} N={N} {np.dtype(dtype).name}")


def test_fused_throughput_mode_and_stage_api_interleave():
    """KEEP_ITERATING (bench mode) and stage calls between fused iterations: ilqr_backward after a fused iteration has
    no expansion in HBM and must produce it first; ILQR_LIN likewise."""
    p = problems.ua_double_pendulum(N=60)
    x0, U0 = problems.ua_batch(48, seed=9, restarts=True, N=60)
    hf, hm = _pair(p, x0, U0, np.float32, tol=p["tol"], maxiter=1 << 30, flags=_lib.FLAG_KEEP_ITERATING)
    for h in (hf, hm):
        h.initial_rollout()
        h.iterate(4)
    _same(hf, hm, "4 iterations")
    # ILQR_LIN is the expansion of the last linearisation: the materialised iterate() leaves the one it swept over (the
    # trajectory before its rollout), the fused one leaves none and produces it on demand at the current trajectory
    lin_f = hf.get(_lib.LIN)
    assert np.isfinite(lin_f).all()
    for h in (hf, hm):
        h.linearize()
    assert np.array_equal(hf.get(_lib.LIN), hm.get(_lib.LIN))
    assert np.array_equal(lin_f, hf.get(_lib.LIN))
    for h in (hf, hm):
        h.backward()
        h.iterate(2)
        h.linearize()
        h.backward()
        h.forward([1.0, 0.5, 0.25])
        h.select()
        h.iterate(1)
    _same(hf, hm, "stage calls interleaved")


def test_fused_mpc_matches_materialised_mpc():
    """The device-resident MPC loop (ilqr_mpc_run): every step's solve ends with the pending acceptance step flushed
    before the plant advances."""
    p = problems.ua_double_pendulum(N=50)
    x0, U0 = problems.ua_batch(32, seed=2, N=50)
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], np.float64)
    out = []
    for flags in (0, _lib.FLAG_NO_FUSE):
        for maxiter in (6, 20):      # enqueue-all form and the counted loop
            h = sysm.make_handle(horizon=50, batch=32, n_alpha=10, n_trials=10, tol=p["tol"], maxiter=maxiter,
                                 plant_integrator="backward_euler", flags=flags)
            h.mpc_reset(x0, U0)
            out.append(h.mpc_run(5))
    for a, b in ((out[0], out[2]), (out[1], out[3])):
        for q in range(3):
            assert np.array_equal(a[q], b[q])
