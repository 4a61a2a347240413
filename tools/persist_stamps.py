"""Cycles per phase of the persistent kernel (workgroup 7, wave 0): diagnostic build
tools/build_variants.sh pstamps "-DILQR_PERSIST_STAMPS -mllvm -amdgpu-sched-strategy=max-ilp"; ILQR_LIB=... ILQR_CLOCK_PROBE=1."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib, problems
p = problems.ua_double_pendulum()
for B in (4096, 1024):
    x0, U0 = problems.ua_batch(B, seed=0)
    h = ilqr_amd.make_system(p["dynamics"], p["cost"], np.float32).make_handle(horizon=200, batch=B, n_alpha=10, maxiter=1 << 30, flags=_lib.FLAG_KEEP_ITERATING)
    h.set_problem(x0, U0); h.initial_rollout(); h.iterate(3); h.iterate(10); h.sync()
    buf = (C.c_longlong * 8)()
    lib = _lib.load()
    lib.ilqr_debug_probe_dump.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_size_t]
    assert lib.ilqr_debug_probe_dump(h.h, buf, 8) == 0
    print(f"B={B}: per iteration (cycles): head+barrier {buf[4] / 10:.0f}, linearise+sweep {buf[5] / 10:.0f}, rollouts {buf[6] / 10:.0f}, tail {buf[7]:.0f}", flush=True)
    h.close()
