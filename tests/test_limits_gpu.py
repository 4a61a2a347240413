"""The hot kernels address X, U, the gains and the tiles through 32-bit buffer descriptors and drop the stores of dead
lanes by pointing them at offset 0x7ffffff0, beyond the descriptor's range (csrc/kernels.hpp forward_ring_kernel,
csrc/backward_tile16.hpp, csrc/backward_fused16.hpp; tools/micro/range_probe.hip measured the rule).  That only holds
while every tensor is at most 0x7ffffff0 bytes: the host's `fits` tests admit exactly that and route anything larger to
the flat-addressed kernels.  This test runs ONE iteration at the largest batch the descriptor path admits for the c3
system -- X = 11 slots x 201 x 16 B x B just below 0x7ffffff0 (the largest odd batch: 60703), so the last wave
and the last workgroup have dead lanes, some trajectories finished -- and compares both ends of the batch and the
neighbours of the finished ones, bit for bit, with a small handle given the same trajectories."""
import numpy as np
import pytest

import ilqr_amd
from ilqr_amd import _lib, problems

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("flags", [0, _lib.FLAG_NO_FUSE])
def test_largest_descriptor_addressed_batch(flags):
    N, n_alpha = 200, 10
    per_traj = (n_alpha + 1) * (N + 1) * 4 * 4            # bytes of X per trajectory (fp32)
    B = 0x7ffffff0 // per_traj - 1                        # 60703: the largest ODD batch the descriptor path admits
    assert B * per_traj <= 0x7ffffff0 < (B + 2) * per_traj and B % 64 != 0 and B % 16 != 0 and B % 4 != 0
    p = problems.ua_double_pendulum(N=N)
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], np.float32)
    pick = np.r_[0:48, B - 47:B]                          # first and last trajectories (the last wave has 31 live lanes)
    x0s, U0s = problems.ua_batch(len(pick), seed=11, restarts=True, N=N)
    x0 = np.zeros((B, 4), np.float32)
    U0 = np.zeros((B, 1, N), np.float32)
    x0[:] = x0s[np.arange(B) % len(pick)]                 # every trajectory is a copy of one of the sampled ones
    U0[:] = U0s[np.arange(B) % len(pick)]
    x0[pick], U0[pick] = x0s, U0s
    big = sysm.make_handle(horizon=N, batch=B, n_alpha=n_alpha, n_trials=10, tol=5.0, maxiter=50, flags=flags)
    small = sysm.make_handle(horizon=N, batch=len(pick), n_alpha=n_alpha, n_trials=10, tol=5.0, maxiter=50, flags=flags)
    big.set_problem(x0, U0)
    small.set_problem(x0s, U0s)
    # iterate the small handle until some -- not all -- of its trajectories have left the loop (tol = 5: dead lanes
    # inside live waves and workgroups), then one more iteration; the big handle runs the same number
    small.initial_rollout()
    n_it = 0
    for n_it in range(1, 40):
        small.iterate(1)
        st = small.get(_lib.STATUS) & 0xff
        if (st != _lib.TRAJ_ACTIVE).any():
            break
    small.iterate(1)
    n_it += 1
    st = small.get(_lib.STATUS) & 0xff
    assert (st == _lib.TRAJ_ACTIVE).any() and (st != _lib.TRAJ_ACTIVE).any(), "pick another tolerance: the sample must be mixed"
    big.initial_rollout()
    big.iterate(n_it)
    if flags == 0:
        # the fused kernel is the same code at both sizes: every bit
        for f in (_lib.COST, _lib.STATUS, _lib.ITERS, _lib.ALPHA, _lib.X, _lib.U, _lib.K, _lib.UFF):
            assert np.array_equal(big.get(f)[pick], small.get(f)), f
    else:
        # materialised: at this batch the tile tensor (2.3 GB) exceeds the descriptor range, so the sweep is the
        # flat-addressed LDS-ring kernel with the transposing step -- same mathematics, last-bit differences against the
        # register-ring kernel of the small handle
        for f in (_lib.STATUS, _lib.ITERS, _lib.ALPHA):
            assert np.array_equal(big.get(f)[pick], small.get(f)), f
        np.testing.assert_allclose(big.get(_lib.COST)[pick], small.get(_lib.COST), rtol=1e-5)
        for f in (_lib.X, _lib.U, _lib.K, _lib.UFF):
            a, b = big.get(f)[pick].astype(np.float64), small.get(f).astype(np.float64)
            assert np.abs(a - b).max() <= 1e-3 * max(1.0, np.abs(b).max()), f
    big.close()
    small.close()
