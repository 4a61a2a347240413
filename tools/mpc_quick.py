"""c4 shard (1024 MPC instances, N=200, maxiter 50): ms per MPC step and the step's attribution (bench.py's mpc_extra)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib, problems
import bench
for dt in (np.float32,):
    r = bench.mpc_extra(ilqr_amd, _lib, problems, dt, 0, None, B=int(os.environ.get("MPC_B", "1024")), n_sim=int(os.environ.get("MPC_STEPS", "10")))
    print(json.dumps(r))
