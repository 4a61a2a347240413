"""Where a sweep wave of the fused kernel spends its cycles (diagnostic build: tools/build_variants.sh fstamps
"-DILQR_FUSED_STAMPS -mllvm -amdgpu-sched-strategy=max-ilp"; run with ILQR_LIB=tools/variants/libilqr_fstamps.so ILQR_CLOCK_PROBE=1)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib, problems
DP = len(sys.argv) > 1 and sys.argv[1] == "dp"      # the fully actuated double pendulum (n=4, m=2), N = 100
p = problems.double_pendulum(N=100) if DP else problems.ua_double_pendulum()
NH = 100 if DP else 200
for B in (4096, 1024):
    if DP:
        rng = np.random.default_rng(0)
        x0, U0 = np.asarray(p["x0"])[None] + 0.1 * rng.standard_normal((B, 4)), np.zeros((B, 2, NH))
    else:
        x0, U0 = problems.ua_batch(B, seed=0)
    h = ilqr_amd.make_system(p["dynamics"], p["cost"], np.float32).make_handle(horizon=NH, batch=B, n_alpha=10, maxiter=1 << 30,
                                                                              flags=_lib.FLAG_KEEP_ITERATING | _lib.FLAG_NO_PERSIST)
    h.set_problem(x0, U0); h.initial_rollout(); h.iterate(6); h.sync()
    buf = (C.c_longlong * 8)()
    lib = _lib.load()
    lib.ilqr_debug_probe_dump.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_size_t]
    assert lib.ilqr_debug_probe_dump(h.h, buf, 8) == 0
    print(f"B={B}: first unit ready after {buf[5]} cycles; waiting for later units {buf[6]} cycles; sweep {buf[7]} cycles = {buf[7] / NH:.0f} per step"
          f" ({(buf[7] - buf[6]) / NH:.0f} without the waits)", flush=True)
    h.close()
