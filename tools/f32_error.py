"""Diagnostic: error of the fp32 mode against the fp64 oracle on c2/c3-shaped inputs (K, k of one
backward sweep; cost of a rollout; full-solve cost), to document why parity is claimed in fp64."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import problems
from oracle.c_oracle import COracle

p = problems.ua_double_pendulum(N=200)
B = 64
x0, U0 = problems.ua_batch(B, seed=1000)
co = COracle(p["dynamics"], p["cost"])
for dt in (np.float32, np.float64):
    s = ilqr_amd.iLQR(ilqr_amd.make_system(p["dynamics"], p["cost"], dt), None, x0, U0, N=200, maxiter=8, verbose=False)
    X, U, c = s.forward_pass(x0, 0.0, np.zeros((B, 4, 201)), U0, np.zeros((B, 1, 200)), np.zeros((B, 200, 1, 4)))
    uff, K = s.backward_pass(X, U)
    eK, ek, ec = [], [], []
    for b in range(B):
        Xo, Uo, c_o = co.forward_pass(x0[b], 0.0, np.zeros((4, 201)), U0[b], np.zeros((1, 200)), np.zeros((200, 1, 4)))
        uff_o, K_o = co.backward_pass(np.asarray(X[b], np.float64), np.asarray(U[b], np.float64))
        eK.append(np.abs(K[b] - K_o).max() / np.abs(K_o).max())
        ek.append(np.abs(uff[b] - uff_o).max() / np.abs(uff_o).max())
        ec.append(abs(c[b] - c_o) / abs(c_o))
    Xs, Us, cs = s.optimize_trajectory()
    es = []
    for b in range(8):
        r = co.solve(x0[b], U0[b], maxiter=8)
        es.append(abs(cs[b] - r["cost"]) / abs(r["cost"]))
    print(np.dtype(dt).name, "max rel err: K %.2e  k %.2e  rollout cost %.2e  solve cost (8 iters) %.2e" %
          (max(eK), max(ek), max(ec), max(es)))
