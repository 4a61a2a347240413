"""Diagnostic: per-workgroup start/end ticks and placement of the backward / forward kernels.
Run on the GPU box with ILQR_CLOCK_PROBE=1."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib, problems
dt = np.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else np.float64
p = problems.ua_double_pendulum(); B = 4096
x0, U0 = problems.ua_batch(B, seed=1000)
s = ilqr_amd.make_system(p["dynamics"], p["cost"], dt)
h = s.make_handle(horizon=200, batch=B, n_alpha=10, maxiter=1 << 30, flags=_lib.FLAG_KEEP_ITERATING)
h.set_problem(x0, U0); h.initial_rollout(); h.iterate(5); h.sync()
n = 8 + 2 * 65536 * 4
buf = np.zeros(n, dtype=np.int64)
h.lib.ilqr_debug_probe_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert h.lib.ilqr_debug_probe_dump(h.h, buf.ctypes.data_as(C.c_void_p), n) == 0
import os
nbw = B // 4 if os.environ.get("ILQR_BACKWARD_LDS_RING") else B // 16
for name, slot, nb in (("backward", 0, nbw), ("forward", 1, (B // 64) * 10)):
    q = buf[8 + slot * 65536 * 4: 8 + slot * 65536 * 4 + nb * 4].reshape(nb, 4)
    t0 = q[:, 0].min()
    st, en, cyc = (q[:, 0] - t0) / 100.0, (q[:, 1] - t0) / 100.0, q[:, 2]
    hw = q[:, 3] & 0xffffffff; xcc = q[:, 3] >> 32
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 3; wave = hw & 0xf
    place = xcc * 1000000 + se * 10000 + sh * 1000 + cu * 10 + simd
    uniq, cnt = np.unique(place, return_counts=True)
    print(f"{name}: {nb} workgroups; start us min/med/max {st.min():.1f}/{np.median(st):.1f}/{st.max():.1f}; "
          f"end us min/med/max {en.min():.1f}/{np.median(en):.1f}/{en.max():.1f}; dur us min/med/max "
          f"{(en-st).min():.1f}/{np.median(en-st):.1f}/{(en-st).max():.1f}; cycles med {np.median(cyc):.0f}; "
          f"distinct (CU,SIMD) of wave 0: {len(uniq)}, hist {np.bincount(cnt).tolist()[:6]}; xcc hist {np.bincount(xcc).tolist()}")
