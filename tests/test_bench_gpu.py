"""bench.py as the driver runs it, on the one GPU of a test box: the headline line carries `roofline`, the
materialised path beside the fused one and the per-config extras; `--config c4` / `--config c5` (the launchers an
8-GPU node runs under torch.distributed.run) run with ONE rank over a one-rank RCCL group; and
ShardedBatch.mpc_run reproduces the un-sharded device-resident MPC loop bit for bit."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import ilqr_amd
from ilqr_amd import problems

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, timeout=300):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout,
                       env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_headline_line_small():
    d = _bench("--steps", "4", "--warmup", "2", "--batch", "2048", "--no-cpu-baseline")     # (> 1024: the two-launch fused path)
    assert d["unit"] == "iLQR iterations/sec" and d["n_gpus"] == 1 and d["steps"] == 4 and d["all_costs_finite"]
    assert abs(d["value"] - 2048 * 4 / (d["ms_per_step"] * 4e-3)) / d["value"] < 1e-9
    assert "fused" in d["config"]["iteration_path"]
    r = d["roofline"]
    assert r["bound"] == "valu" and r["avg_launch_us"] > 0 and r["launches"] == 4
    rm = d["roofline_materialised"]
    assert rm["bound"] == "hbm" and 0 < rm["frac"] < 1 and rm["launches"] == 4
    assert set(d["materialised"]["kernels"]) >= {"linearize_kernel", "backward_tile16_kernel", "forward_ring_kernel"}
    assert set(d["kernels"]) >= {"backward_fused16_kernel", "forward_ring_kernel"}
    for tag in ("c1", "c2", "c3_f64", "dp_4x2", "c5_shard", "c5"):
        c = d["configs"][tag]
        assert c["ms_per_iteration"] > 0 and c["kernels"], tag
    m = d["mpc_c4_shard"]
    assert m["all_finite"] and m["host_looped"]["same_result"] and m["host_looped"]["attribution"]["iteration_launches_per_step"] >= 1
    s = d["solve_to_convergence"]
    assert s["converged"] + s["linesearch_failed"] + s["maxiter"] == 2048


def test_headline_under_torch_distributed_run_one_rank():
    """The launcher form the driver uses for N > 1 (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`), with
    N = 1: RANK / WORLD_SIZE / MASTER_* from the environment, the side-stream status exchange over a one-rank RCCL group."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--batch", "256", "--no-cpu-baseline", "--no-solve-extra", "--exchange", "--exchange-every", "1"],
                       capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and "exchange" in d and d["all_costs_finite"]


@pytest.mark.parametrize("extra,unit,total", [(["--batch", "256", "--exchange-every", "1", "--no-cpu-baseline", "--no-solve-extra"], "iLQR iterations/sec", 512),
                                               (["--config", "c4", "--batch", "64"], "MPC instance-steps/sec", 64),
                                               (["--config", "c5", "--batch", "32"], "iLQR iterations/sec", 32)])
def test_two_ranks_rehearsed_on_one_card(extra, unit, total):
    """The N > 1 control flow of bench.py -- ranks from the launcher's environment, per-rank shards and seeds, the status
    exchange between iterations, barrier + MAX of the wall time over ranks, one JSON line from rank 0 -- with TWO ranks
    sharing the one GPU of a test box.  RCCL refuses two ranks on one device, so the collectives run over gloo on host
    copies (`--backend gloo`); RCCL itself is covered by the one-rank tests.  What an 8-GPU node adds is one rank per card."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--backend", "gloo", *extra], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["unit"] == unit and d["steps"] == 3
    assert abs(d["value"] - total * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-9
    if "--config" not in extra:
        assert d["scaling"] == "weak" and "2 independent shards" in d["config"]["sharding"]
    else:
        assert d["config"].get("instances_per_gpu", d["config"].get("batch_per_gpu")) == total // 2


def test_headline_small_shard_runs_persistent():
    """A shard of <= 1024 trajectories: the whole iteration is one launch of ilqr_persistent_kernel, and the line says so."""
    d = _bench("--steps", "3", "--warmup", "1", "--batch", "256", "--no-cpu-baseline", "--no-solve-extra")
    assert "persistent" in d["config"]["iteration_path"] and "ilqr_persistent_kernel" in d["roofline"]["kernel"]
    assert d["roofline"]["launches"] == 3 and d["roofline"]["avg_launch_us"] > 0 and d["all_costs_finite"]
    assert any(k.startswith("ilqr_persistent_kernel") for k in d["kernels"])


def test_headline_materialised_switch():
    d = _bench("--steps", "3", "--warmup", "1", "--batch", "256", "--no-cpu-baseline", "--no-solve-extra", "--materialised")
    assert "materialised" in d["config"]["iteration_path"] and d["roofline"]["bound"] == "hbm"


@pytest.mark.parametrize("cfg,batch,unit", [("c4", "96", "MPC instance-steps/sec"), ("c5", "48", "iLQR iterations/sec")])
def test_sharded_configs_one_rank(cfg, batch, unit):
    d = _bench("--config", cfg, "--batch", batch, "--steps", "2", "--warmup", "1")
    assert d["unit"] == unit and d["n_gpus"] == 1 and d["value"] > 0
    assert d["config"].get("instances_per_gpu", d["config"].get("batch_per_gpu")) == int(batch)


def test_sharded_mpc_equals_unsharded():
    import torch
    from ilqr_amd import dist as idist
    p = problems.ua_double_pendulum(N=40)
    B = 40
    x0, U0 = problems.ua_batch(B, seed=2, N=40)
    plant = ilqr_amd.make_system(dict(p["dynamics"], integrator=p["plant_integrator"]), p["cost"])
    with torch.cuda.stream(torch.cuda.Stream()):
        sb = idist.ShardedBatch(lambda: ilqr_amd.make_system(p["dynamics"], p["cost"]), x0, U0, N=40, tol=p["tol"], maxiter=15,
                                plant=plant)
        sb.mpc_reset()
        u, x, c = sb.mpc_run(4)
        st = sb.global_status()
    ref = ilqr_amd.iLQR(ilqr_amd.make_system(p["dynamics"], p["cost"]), None, x0, U0, N=40, tol=p["tol"], maxiter=15, plant=plant,
                        verbose=False)
    ref.mpc_reset(x0, U0)
    ur, xr, cr = ref.mpc_run(4)
    assert np.array_equal(u, ur) and np.array_equal(x, xr) and np.array_equal(c, cr)
    assert st.n_active == 0 and abs(st.min_cost - float(np.min(c[-1]))) <= 1e-12 * abs(st.min_cost)
