"""Which predecessor makes the (16, 8) sweep slow inside an iteration at B = 1024?  (profiles/r02/c5_sweep.log: 618 us back
to back, 1029 us in the iteration.)  Times backward_mfma16_kernel (per-dispatch HIP events) behind different kernels."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib, problems

p = problems.linear_quadratic()
B = int(os.environ.get("C5_B", "1024"))
x0, U0 = problems.lq_batch(B, 16, 8, 500)
h = ilqr_amd.make_system(p["dynamics"], p["cost"], np.float32).make_handle(horizon=500, batch=B, n_alpha=10, maxiter=1 << 30,
                                                                          flags=_lib.FLAG_KEEP_ITERATING)
h.set_problem(x0, U0); h.initial_rollout(); h.iterate(2); h.sync()
alphas = [2.0 ** -k for k in range(10)]


def run(name, seq, reps=6):
    h.timing_enable(True); h.timing_reset()
    for _ in range(reps):
        for s in seq:
            if s == "L": h.linearize()
            elif s == "B": h.backward()
            elif s == "F": h.forward(alphas)
            elif s == "S": h.select()
    t = h.timing_get()
    print(f"{name:28s}", {k: round(v[0] / v[1] * 1e3, 1) for k, v in t.items() if v[1]}, flush=True)
    h.timing_enable(False)


run("B B B ...", "B")
run("L B L B ...", "LB")
run("F S B ...", "FSB")
run("L B F S (iteration)", "LBFS")
run("B B B ... again", "B")
