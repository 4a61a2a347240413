"""Cycles the fp32 tile sweep spends in its ring wait (diagnostic build: tools/build_variants.sh t16stamps
"-DILQR_T16_STAMPS"; run with ILQR_LIB=tools/variants/libilqr_t16stamps.so ILQR_CLOCK_PROBE=1).  Every wait is
bracketed by two s_memtime reads and preceded by an empty pair, so wait - calibration = cycles stalled in s_waitcnt."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib, problems
p = problems.ua_double_pendulum()
lib = _lib.load()
lib.ilqr_debug_probe_dump.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_size_t]
for B in (512, 4096):
    x0, U0 = problems.ua_batch(B, seed=0)
    h = ilqr_amd.make_system(p["dynamics"], p["cost"], np.float32).make_handle(horizon=200, batch=B, n_alpha=10, maxiter=1 << 30,
                                                                             flags=_lib.FLAG_KEEP_ITERATING)
    h.set_problem(x0, U0); h.initial_rollout(); h.iterate(2)
    for mode in ("in iteration", "back to back"):
        if mode == "in iteration":
            h.linearize(); h.backward()
        else:
            h.backward(); h.backward()
        h.sync()
        buf = (C.c_longlong * 24)()
        assert lib.ilqr_debug_probe_dump(h.h, buf, 24) == 0
        wait, cal, tot = buf[2], buf[3], buf[4]
        n = 190
        print(f"B={B} {mode}: total {tot / 200:.0f} cycles/step (stamped build), in wait {(wait - cal) / n:.0f} cycles/step "
              f"(raw {wait / n:.0f}, empty pair {cal / n:.0f})")
    h.close()
