"""Experiment: CU-masked streams.  (1) two half-batch solvers on disjoint halves of the CUs, iterating concurrently;
(2) phase partition: S sub-batches, rollouts on streams masked to one CU set, linearise/sweep/select on streams masked
to the other, chained with events."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ilqr_amd
from ilqr_amd import _lib, problems

hip = C.CDLL("libamdhip64.so")
lib = _lib.load()
lib.ilqr_debug_set_stream.argtypes = [C.c_void_p, C.c_void_p]

def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[int(sum(1 << k for k in range(32) if bits[32 * w + k])) for w in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return s

def event():
    e = C.c_void_p(); assert hip.hipEventCreateWithFlags(C.byref(e), 2) == 0; return e   # hipEventDisableTiming

p = problems.ua_double_pendulum()
B, N = 4096, 200
x0, U0 = problems.ua_batch(B, seed=0)
sysm = ilqr_amd.make_system(p["dynamics"], p["cost"], np.float32)
alphas = (C.c_double * 10)(*[0.5 ** k for k in range(10)])

def make(lo, hi, stream):
    h = sysm.make_handle(horizon=N, batch=hi - lo, n_alpha=10, maxiter=1 << 30, flags=_lib.FLAG_KEEP_ITERATING, stream=stream)
    h.set_problem(x0[lo:hi], U0[lo:hi]); h.initial_rollout(); h.iterate(3); h.sync()
    return h

allcu = [True] * 256
# ---- (0) reference: one handle, unmasked ------------------------------------------------------------------
h = make(0, B, masked_stream(allcu))
t0 = time.perf_counter(); h.iterate(20); h.sync(); print(f"one handle, all CUs: {(time.perf_counter()-t0)/20*1e6:.0f} us/iter", flush=True)
h.close()
# ---- (1) isolation: two halves on disjoint CU halves --------------------------------------------------------
for name, m0, m1 in (("low/high bit halves", [i < 128 for i in range(256)], [i >= 128 for i in range(256)]),
                     ("even/odd bits", [i % 2 == 0 for i in range(256)], [i % 2 == 1 for i in range(256)]),
                     ("unmasked", allcu, allcu)):
    hs = [make(0, B // 2, masked_stream(m0)), make(B // 2, B, masked_stream(m1))]
    t0 = time.perf_counter()
    for h in hs: h.iterate(20)
    for h in hs: h.sync()
    both = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter(); hs[0].iterate(20); hs[0].sync(); alone = (time.perf_counter() - t0) / 20
    print(f"two halves, {name}: both {both*1e6:.0f} us per joint iteration; first alone on its mask {alone*1e6:.0f} us", flush=True)
    for h in hs: h.close()
# ---- (2) phase partition ----------------------------------------------------------------------------------
def phase_partition(S, n_fwd_cus):
    fmask = [i < n_fwd_cus for i in range(256)]
    lmask = [i >= n_fwd_cus for i in range(256)]
    edges = np.linspace(0, B, S + 1).astype(int)
    subs = []
    for s in range(S):
        sf, sl = masked_stream(fmask), masked_stream(lmask)
        hdl = make(edges[s], edges[s + 1], sl)
        subs.append((hdl, sf, sl, event(), event()))
    def run(iters):
        for k in range(iters):
            for (hdl, sf, sl, e_l, e_f) in subs:
                lib.ilqr_debug_set_stream(hdl.h, sl)
                hip.hipStreamWaitEvent(sl, e_f, 0)          # previous rollout of this sub-batch (no-op before the first record)
                hdl._chk(lib.ilqr_select(hdl.h)) if k else None
                hdl._chk(lib.ilqr_linearize(hdl.h)); hdl._chk(lib.ilqr_backward(hdl.h))
                hip.hipEventRecord(e_l, sl)
                lib.ilqr_debug_set_stream(hdl.h, sf)
                hip.hipStreamWaitEvent(sf, e_l, 0)
                hdl._chk(lib.ilqr_forward(hdl.h, alphas, 10))
                hip.hipEventRecord(e_f, sf)
        for (hdl, sf, sl, e_l, e_f) in subs:
            lib.ilqr_debug_set_stream(hdl.h, sl); hip.hipStreamWaitEvent(sl, e_f, 0); hdl._chk(lib.ilqr_select(hdl.h))
        for (hdl, sf, sl, _, _) in subs:
            hip.hipStreamSynchronize(sf); hip.hipStreamSynchronize(sl)
    run(3)
    t0 = time.perf_counter(); run(20); el = (time.perf_counter() - t0) / 20
    print(f"phase partition S={S}, rollout CUs {n_fwd_cus}: {el*1e6:.0f} us per whole-batch iteration = {B/el/1e6:.2f} M it/s", flush=True)
    for (hdl, *_ ) in subs: hdl.close()
for S, nf in ((2, 96), (2, 128), (3, 128), (3, 160), (4, 128), (4, 160), (1, 160)):
    phase_partition(S, nf)
