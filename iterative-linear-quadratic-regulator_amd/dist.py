"""Multi-GPU sharding of the batched solve: one process per GPU, each owning a
contiguous shard of independent trajectories; no trajectory data ever crosses GPUs.  The only
exchange is a scalar all-reduce of {min cost, max |dcost|, #active, #converged} so that every rank
knows the best cost and whether the whole job has converged -- RCCL over xGMI on the GPUs
(``torch.distributed`` backend "nccl"), gloo in the CPU tests.

The reference has no counterpart (it is a single process); the semantics follow its loop
(iLQR_class.py:267, 304-311) applied to the union of all shards.
"""

from __future__ import annotations

from dataclasses import dataclass


def shard_range(total: int, world: int, rank: int):
    """Contiguous [lo, hi) of `total` trajectories owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


@dataclass
class GlobalStatus:
    min_cost: float
    max_dcost: float
    n_active: int
    n_converged: int


def allreduce_status(stats4, group=None):
    """In-place all-reduce of a 4-vector {min cost, max |dcost|, #active, #converged} (float64 tensor,
    on the GPU for nccl/RCCL or on the CPU for gloo): MIN, MAX, SUM, SUM -- as two collectives
    (-min and max share one MAX)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return stats4      # no process group: a single shard (with a group of ONE rank the collectives still run)
    if stats4.is_cuda and dist.get_backend(group) == "gloo":
        # several ranks rehearsed on ONE card (tests, `bench.py --backend gloo`): the collective runs on a host copy
        host = allreduce_status(stats4.cpu(), group)
        stats4.copy_(host)
        return stats4
    mm = torch.stack([-stats4[0], stats4[1]])
    dist.all_reduce(mm, op=dist.ReduceOp.MAX, group=group)
    cnt = stats4[2:4].clone()
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
    stats4[0] = -mm[0]
    stats4[1] = mm[1]
    stats4[2:4] = cnt
    return stats4


def to_status(stats4) -> GlobalStatus:
    v = [float(x) for x in stats4.tolist()]
    return GlobalStatus(min_cost=v[0], max_dcost=v[1], n_active=int(round(v[2])), n_converged=int(round(v[3])))


def local_stats(cost, cost_prev, status):
    """Host-side twin of the device kernel status_reduce_kernel (csrc/kernels.hpp), used by the CPU
    tests and by callers that already hold host copies: returns a float64 CPU tensor of 4."""
    import numpy as np
    import torch
    cost = np.asarray(cost, dtype=np.float64)
    cost_prev = np.asarray(cost_prev, dtype=np.float64)
    st = np.asarray(status) & 0xff
    return torch.tensor([cost.min(), np.abs(cost - cost_prev).max(), float((st == 0).sum()), float((st == 1).sum())],
                        dtype=torch.float64)


class StatusExchange:
    """The per-iteration inter-GPU exchange, off the critical path: each rank's 4-vector
    {min cost, max |dcost|, #active, #converged} is all-gathered (ONE collective of 32 B per rank) on a side
    stream while the compute stream goes straight on to the next iteration; the host reduces the gathered
    rows when it wants the global status.  Double-buffered, so a launch never overwrites the operand of a
    collective that may still be reading it.  Works on CUDA tensors (nccl = RCCL over xGMI) and on CPU
    tensors (gloo, used by the tests; no streams there)."""

    def __init__(self, device=None, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.cuda = device is not None and str(device) != "cpu"
        dev = device if self.cuda else "cpu"
        self.stats = [torch.zeros(4, dtype=torch.float64, device=dev) for _ in range(2)]
        self.gathered = [torch.zeros(self.world, 4, dtype=torch.float64, device=dev) for _ in range(2)]
        self.k = 0
        # gloo with device tensors (several ranks rehearsed on one card): the gather runs on host copies, blocking
        self.host_gather = self.cuda and dist.is_initialized() and dist.get_backend(group) == "gloo"
        if self.host_gather:
            self.gathered_host = [torch.zeros(self.world, 4, dtype=torch.float64) for _ in range(2)]
        if self.cuda:
            self.side = torch.cuda.Stream(device=dev)
            # one event pair per buffer, re-recorded at every use (creating events per step is host time on a 0.2 ms step)
            self.ready = [torch.cuda.Event(), torch.cuda.Event()]
            self.done = [torch.cuda.Event(), torch.cuda.Event()]
            self.used = [False, False]

    def launch(self, fill):
        """fill(stats_tensor) writes this rank's 4-vector ON TORCH'S CURRENT STREAM (e.g. handle.status_reduce of a
        handle created with stream=torch.cuda.current_stream().cuda_stream, as ShardedBatch and bench.py do): the
        side-stream collective is ordered behind an event recorded on that stream.  A handle that owns a private
        stream must be synchronised inside `fill` (handle.sync()) before it returns."""
        torch, dist = self.torch, self.dist
        i = self.k & 1
        self.k += 1
        if self.host_gather:
            fill(self.stats[i])
            torch.cuda.current_stream().synchronize()
            dist.all_gather(list(self.gathered_host[i].unbind(0)), self.stats[i].cpu(), group=self.group)
            self.gathered[i].copy_(self.gathered_host[i])
            return i
        if self.cuda:
            cur = torch.cuda.current_stream()
            # the collective that last used this buffer pair (two launches ago) must have finished: normally it has, and
            # then the compute stream is spared the wait packet
            if self.used[i] and not self.done[i].query():
                cur.wait_event(self.done[i])
            fill(self.stats[i])
            self.ready[i].record(cur)
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.ready[i])
                self._gather(i)
                self.done[i].record(self.side)
            self.used[i] = True
        else:
            fill(self.stats[i])
            self._gather(i)
        return i

    def _gather(self, i):
        if not self.dist.is_initialized():
            self.gathered[i][0].copy_(self.stats[i])          # no process group at all: a single shard
        else:                                                  # (a group of ONE rank still goes through RCCL / gloo)
            self.dist.all_gather(list(self.gathered[i].unbind(0)), self.stats[i], group=self.group)

    def result(self, i=None) -> GlobalStatus:
        if i is None:
            i = (self.k - 1) & 1
        if self.cuda:
            self.side.synchronize()
        g = self.gathered[i].cpu()
        return GlobalStatus(min_cost=float(g[:, 0].min()), max_dcost=float(g[:, 1].max()),
                            n_active=int(round(float(g[:, 2].sum()))), n_converged=int(round(float(g[:, 3].sum()))))


class ShardedBatch:
    """Splits a global batch (x0 (B, n), U_init (B, m, N)) over the ranks of the default process group
    and solves the local shard with ``iLQR``; ``global_status()`` is the RCCL all-reduce."""

    def __init__(self, system_factory, x0, U_init, device=None, **ilqr_kw):
        import torch
        import torch.distributed as dist
        from .iLQR_class import iLQR
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.lo, self.hi = shard_range(len(x0), self.world, self.rank)
        if device is None:
            device = torch.cuda.current_device()
        # the handle launches on torch's current stream, so torch events / collectives are ordered with its kernels
        ilqr_kw.setdefault("stream", torch.cuda.current_stream(device).cuda_stream or None)
        self._x0, self._U0 = x0[self.lo:self.hi], U_init[self.lo:self.hi]
        self.solver = iLQR(system_factory(), None, self._x0, self._U0, device=device, verbose=False, **ilqr_kw)
        self._stats = torch.zeros(4, dtype=torch.float64, device=f"cuda:{device}")
        self._xchg = None

    def solve(self):
        X, U, cost = self.solver.optimize_trajectory()
        return X, U, cost

    def mpc_reset(self, keep_state=False):
        """Start the shard's device-resident controllers at the plant states / warm starts given to the constructor
        (run_iLQR_UA_MPC.py:146-174 for every instance of the shard; c4 = 8192 instances, 1024 per GPU)."""
        self.solver.mpc_reset(self._x0, self._U0, keep_state=keep_state)

    def mpc_run(self, n_steps):
        """n_steps receding-horizon steps of every instance of this rank's shard, device-resident (ilqr_mpc_run).
        No data crosses GPUs; call global_status() for the all-reduced {min cost, max |dcost|, #active, #converged}.
        Returns this shard's (U_sim, X_sim, cost) = (n_steps, B_local, n_u), (n_steps, B_local, n_x), (n_steps, B_local)."""
        return self.solver.mpc_run(n_steps)

    def global_status(self) -> GlobalStatus:
        import torch
        h = self.solver.handle
        h.status_reduce(self._stats.data_ptr())
        h.sync()
        torch.cuda.synchronize()
        allreduce_status(self._stats)
        return to_status(self._stats.cpu())

    def global_status_async(self) -> GlobalStatus:
        """The same reduction through the side-stream all-gather (StatusExchange): what bench.py runs per step."""
        import torch
        if self._xchg is None:
            self._xchg = StatusExchange(device=self._stats.device)
        h = self.solver.handle

        def fill(t):
            h.status_reduce(t.data_ptr())
            h.sync()      # also correct for a handle on a private stream

        return self._xchg.result(self._xchg.launch(fill))
