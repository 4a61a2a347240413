"""Phase stamps of the (16, 8) MFMA sweep (diagnostic build: tools/build_variants.sh stamps "-DILQR_MFMA16_STAMPS";
run with ILQR_LIB=tools/variants/libilqr_stamps.so ILQR_CLOCK_PROBE=1)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib, problems
p = problems.linear_quadratic()
x0, U0 = problems.lq_batch(128, 16, 8, 500)
h = ilqr_amd.make_system(p["dynamics"], p["cost"], np.float32).make_handle(horizon=500, batch=128, maxiter=1 << 30, flags=_lib.FLAG_KEEP_ITERATING)
h.set_problem(x0, U0); h.initial_rollout(); h.iterate(2); h.linearize(); h.backward(); h.backward(); h.sync()
buf = (C.c_longlong * 24)()
lib = _lib.load()
lib.ilqr_debug_probe_dump.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_size_t]
assert lib.ilqr_debug_probe_dump(h.h, buf, 24) == 0
names = ["Put, Pt ready", "Quu+Qux ready", "Pt,Qxx,qu,qx ready", "readlane + rhs", "Cholesky", "substitution", "K, V+, gains, Vx", "cur=nxt (tile loads landed)", "next tile's loads issued", "this tile and V available"]
tot = sum(buf[8 + k] for k in range(10))
for k in range(10):
    print(f"{names[k]:32s} {buf[8 + k] / 500:8.0f} cycles/step  {100.0 * buf[8 + k] / tot:5.1f} %")
print(f"{'total':32s} {tot / 500:8.0f} cycles/step (stamped build)")
