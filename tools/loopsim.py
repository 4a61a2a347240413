"""In-order issue model of a kernel's innermost loop, from a `hipcc -S` dump, with the lone-wave costs measured by
tools/micro/issue_rate.hip: an independent vector instruction 4.1 cycles, its result consumable 8.7 cycles after it
issued (12.5 behind a transcendental, which itself holds the port 9), v_readlane 8.75, a vector-memory load 13.5 and a
store 21 of the wave's own issue time, s_nop N = N + 1.  Prints the instruction mix, cycles per trip and how many of
them are dependency stalls -- i.e. what a perfect reordering of the same instructions could win back.

    hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S -o f32.s ilqr_f32.hip      (in csrc/, with the Makefile's flags)
    python tools/loopsim.py f32.s forward_ring_kernelIfNS_14DoublePendulumIfLi1EEELi2E [steps per trip]
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                "iterative-linear-quadratic-regulator_amd", "csrc"))
import verify_ring_isa as v

path, name = sys.argv[1], sys.argv[2]
per_trip = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lines = open(path, errors="replace").read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and l.rstrip().split(":")[0].endswith("E") or
             (l.startswith("_Z") and name in l and ":" in l))
stop = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
txt = lines[start:stop]
hdr = [i for i, l in enumerate(txt) if "Inner Loop Header" in l][-1]
end = next(i for i in range(hdr, len(txt)) if "s_cbranch_scc" in txt[i])
body = [l.strip() for l in txt[hdr + 1:end + 1] if l.strip() and not l.strip().startswith((";", "."))]


def cls(m):
    if m.startswith(("buffer_load", "global_load")): return "load"
    if m.startswith(("buffer_store", "global_store")): return "store"
    if m.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos", "v_exp", "v_log")): return "trans"
    if m.startswith(("v_readlane", "v_readfirstlane")): return "readlane"
    if m.startswith("v_mfma"): return "mfma"
    if m.startswith("v_"): return "valu"
    if m.startswith("s_nop"): return "nop"
    if m.startswith("s_waitcnt"): return "wait"
    return "salu"


COST = {"valu": (4.1, 8.7), "trans": (9.0, 12.5), "readlane": (8.75, 8.75), "mfma": (4.1, 40.0), "load": (13.5, 0.0),
        "store": (21.0, 0.0), "wait": (0.0, 0.0), "salu": (2.0, 0.0)}
t, stall, ready, mix = 0.0, 0.0, {}, {}
for l in body:
    m = l.split()[0]
    ops = l[len(m):]
    c = cls(m)
    mix[c] = mix.get(c, 0) + 1
    fields = [f.strip() for f in ops.split(",")]
    writes = c in ("valu", "trans", "mfma", "load")
    dst = set(v._regs(fields[0])) if writes else set()
    src = set()
    for f in (fields[1:] if writes else fields):
        src |= set(v._regs(f))
    if "fmac" in m or "fmaak" in m:
        src |= dst
    need = max([ready.get(r, 0.0) for r in src] + [0.0])
    if need > t:
        if os.environ.get("LOOPSIM_VERBOSE") and need - t > 3:
            print(f"  stall {need - t:5.1f} at {l}")
        stall += need - t
        t = need
    dur, lat = (int(ops.strip() or 0) + 1, 0.0) if c == "nop" else COST[c]
    if c != "load":
        for r in dst:
            ready[r] = t + lat
    t += dur
print(f"{len(body)} instructions per trip: {mix}")
print(f"modelled cycles per trip {t:.0f} = {t / per_trip:.0f} per step, of which dependency stalls {stall / per_trip:.0f} per step")
