"""The multi-GPU hook on the one GPU a test box has: a process group of ONE rank over the `nccl` backend (= RCCL on
ROCm), so the code that an 8-GPU job runs per step -- ilqr_status_reduce on the handle's stream, the event hand-over to
the side stream, the RCCL all-gather / all-reduce, the host-side reduction (dist.py) -- has executed on hardware before
a multi-GPU node ever sees it.  The N > 1 arithmetic itself is covered by the world-size-2
gloo tests in tests/test_dist_cpu.py."""
import os
import socket

import numpy as np
import pytest

import ilqr_amd
from ilqr_amd import _lib, dist as idist, problems

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl_group():
    import torch
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    yield dist
    dist.destroy_process_group()


def _solve_with_trace(h, iters):
    """cost, cost_prev, status as the device holds them after `iters` iterations (cost_prev is not an ilqr_get field:
    it is the cost before the last ACCEPTED step, select_kernel)."""
    h.initial_rollout()
    cost = h.get(_lib.COST).astype(np.float64)
    cost_prev = cost.copy()
    for _ in range(iters):
        before = cost
        running = (h.get(_lib.STATUS) & 0xff) == 0          # a finished trajectory is frozen, its alpha_taken too
        h.iterate(1)
        cost = h.get(_lib.COST).astype(np.float64)
        took = running & (h.get(_lib.ALPHA) > 0)
        cost_prev = np.where(took, before, cost_prev)
    return cost, cost_prev, h.get(_lib.STATUS)


@pytest.mark.parametrize("tol,iters", [(1e-5, 3), (0.3, 7)])       # everybody still iterating / everybody converged
@pytest.mark.parametrize("own_stream", [False, True])
def test_status_exchange_over_rccl_single_rank(rccl_group, own_stream, tol, iters):
    import torch
    p = problems.ua_double_pendulum(N=40)
    B = 300
    x0, U0 = problems.ua_batch(B, seed=9, restarts=True, N=40)
    x0 = x0 * np.linspace(0.0, 2.0, B)[:, None]
    sysm = ilqr_amd.make_system(p["dynamics"], p["cost"])
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                          # a non-default torch stream is current
        stream = None if own_stream else torch.cuda.current_stream().cuda_stream
        h = sysm.make_handle(horizon=40, batch=B, tol=tol, maxiter=25, stream=stream)
        h.set_problem(x0, U0)
        cost, cost_prev, status = _solve_with_trace(h, iters)
        want = idist.to_status(idist.local_stats(cost, cost_prev, status))
        assert want.n_active + want.n_converged == B and (want.n_active == B or want.n_converged == B)
        xchg = idist.StatusExchange(device="cuda:0")
        assert xchg.world == 1 and xchg.cuda

        def fill(t):
            h.status_reduce(t.data_ptr())
            if own_stream:
                h.sync()                                   # the handle's private stream is not torch's
        for _ in range(3):                                 # both buffers of the double buffer, and their reuse
            got = xchg.result(xchg.launch(fill))
            assert (got.n_active, got.n_converged) == (want.n_active, want.n_converged)
            np.testing.assert_allclose([got.min_cost, got.max_dcost], [want.min_cost, want.max_dcost], rtol=1e-12)
        # the blocking form: one MAX + one SUM all-reduce
        t = torch.zeros(4, dtype=torch.float64, device="cuda:0")
        h.status_reduce(t.data_ptr())
        h.sync()
        torch.cuda.synchronize()
        got = idist.to_status(idist.allreduce_status(t).cpu())
        assert (got.n_active, got.n_converged) == (want.n_active, want.n_converged)
        np.testing.assert_allclose([got.min_cost, got.max_dcost], [want.min_cost, want.max_dcost], rtol=1e-12)
        h.close()


def test_sharded_batch_on_one_rank(rccl_group):
    """ShardedBatch (shard_range + iLQR on torch's current stream + the two reductions) with world size 1."""
    import torch
    p = problems.ua_double_pendulum(N=40)
    B = 96
    x0, U0 = problems.ua_batch(B, seed=4, restarts=True, N=40)
    with torch.cuda.stream(torch.cuda.Stream()):
        sb = idist.ShardedBatch(lambda: ilqr_amd.make_system(p["dynamics"], p["cost"]), x0, U0, N=40, tol=p["tol"], maxiter=5)
        assert (sb.lo, sb.hi, sb.world) == (0, B, 1)
        X, U, cost = sb.solve()
        st = sb.global_status()
        st2 = sb.global_status_async()
    names = np.array(sb.solver.status)
    assert st.n_active == 0 and st.n_converged == int((names == "converged").sum())
    np.testing.assert_allclose(st.min_cost, np.min(cost), rtol=1e-12)
    assert (st2.n_active, st2.n_converged) == (st.n_active, st.n_converged)
    np.testing.assert_allclose([st2.min_cost, st2.max_dcost], [st.min_cost, st.max_dcost], rtol=1e-12)
