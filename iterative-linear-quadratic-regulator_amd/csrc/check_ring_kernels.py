"""Build-time guard for the kernels that count their own memory operations.

forward_ring_kernel and backward_tile16_kernel keep a ring of inline-asm buffer loads in flight and wait for
them with hand-counted ``s_waitcnt vmcnt(N)``.  Between a slot's issue and its wait the compiler believes the
destination registers already hold their values; if register pressure makes it spill them in that window
(to scratch, or to AGPRs with v_accvgpr_write) it copies registers whose loads have not landed, and the
restored values are garbage.  No language-level construct forbids that, so the invariant is checked on the
compiler's own resource report (-Rpass-analysis=kernel-resource-usage): such a kernel must use no AGPRs (hipcc
only touches them to park VGPRs: there is no MFMA here) and report no VGPR spills.  SGPR spills -- to VGPR
lanes, or through them to scratch -- are harmless: they never move a ring register.  The library build fails on a violation; the plugin builder (systems/custom_sys.py) instead
recompiles with the offending integrators routed to the compiler-scheduled forward_kernel.

A spill is only one way to touch an in-flight register; a plain register copy is another (it broke the fp64 sweep with
dropped stores in round 1).  Given the compiler's assembly as a second argument, verify_ring_isa.py walks every path
of the guarded kernels with the vmcnt queue simulated and rejects ANY access to the destination of an asm load that
may still be in flight.

A third check covers EVERY function of the assembly: a 16-byte buffer store whose next instruction is a packed-FP32 write of
its data registers stores the new value in part of the lanes on gfx950 (measured, tools/micro/store_hazard.hip); hipcc pads
that pair only for stores without an SGPR offset.  See verify_ring_isa.store_pk_hazards.

usage: check_ring_kernels.py <hipcc stderr log> [<device .s>]      (exit 1 and a list on violation)
"""
import re
import sys

GUARDED = ("forward_ring_kernel", "backward_tile16_kernel", "backward_tile16m2_kernel", "forward_mfma16_kernel")
# functions (not kernels: the compiler's resource report has no entry for them) that hold a self-counted ring: the
# assembly check covers them -- a spill of an in-flight register is an access to it
GUARDED_FUNCS = ("role_rollout",)


def parse(log_text):
    """-> list of dicts {name, vgprs, agprs, scratch, vspill} for every kernel in a resource-usage log."""
    out, cur = [], None
    for line in log_text.splitlines():
        m = re.search(r"\bFunction Name: (\S+)", line)      # ("file:line:col: remark: ..." or "remark: file:line:col: ...")
        if m:
            cur = {"name": m.group(1), "vgprs": 0, "agprs": 0, "scratch": 0, "vspill": 0}
            out.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("vgprs", r"\s+VGPRs: (\d+)"), ("agprs", r"\s+AGPRs: (\d+)"),
                         ("scratch", r"\s+ScratchSize \[bytes/lane\]: (\d+)"),
                         ("vspill", r"\s+VGPRs Spill: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    return out


# Kernels whose AGPRs are MFMA accumulators, not spill space: the AGPR count says nothing there, and a spill into an
# AGPR (v_accvgpr_write of an in-flight register) is caught by the assembly check instead.
MFMA_KERNELS = ("forward_mfma16_kernel",)


def violations(kernels):
    return [k for k in kernels if any(g in k["name"] for g in GUARDED) and
            ((k["agprs"] and not any(m in k["name"] for m in MFMA_KERNELS)) or k["vspill"])]


def forward_ring_integrator(name):
    """Integrator template argument of a mangled forward_ring_kernel<T, Dyn, INTEG> name, or None."""
    if "forward_ring_kernel" not in name:
        return None
    m = re.search(r"ELi(\d+)EEEvNS_5KArgs", name)
    return int(m.group(1)) if m else None


def isa_violations(asm_path):
    """Kernel names whose assembly touches the destination of an in-flight asm load (verify_ring_isa.py), plus the
    full per-kernel result."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("ilqr_verify_ring_isa",
                                                  os.path.join(os.path.dirname(os.path.abspath(__file__)), "verify_ring_isa.py"))
    vri = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(vri)
    res = vri.verify_text(open(asm_path, errors="replace").read(), GUARDED + GUARDED_FUNCS)
    return [k for k, r in res.items() if r["violations"]], res, vri


def store_hazards(asm_path):
    """(function, .s line, store, follower) for every wide buffer store directly followed by a packed-FP32 write of its data
    registers, anywhere in the file (verify_ring_isa.store_pk_hazards)."""
    return isa_violations(asm_path)[2].store_pk_hazards(open(asm_path, errors="replace").read())


if __name__ == "__main__":
    # usage: check_ring_kernels.py <resource-usage log> [<device .s file>]
    ks = parse(open(sys.argv[1], errors="replace").read())
    guarded = [k for k in ks if any(g in k["name"] for g in GUARDED)]
    bad = violations(ks)
    for k in bad:
        sys.stderr.write(f"ring kernel spills (agprs {k['agprs']}, vgpr spills {k['vspill']}): {k['name']}\n")
    if not guarded:
        sys.stderr.write("check_ring_kernels: no guarded kernel found in the log (was the remark flag passed?)\n")
        sys.exit(1)
    n_isa = 0
    if len(sys.argv) > 2:
        bad_isa, res, vri = isa_violations(sys.argv[2])
        n_isa = vri.report(res)
        n_kern = sum(1 for k in res if not any(g in k for g in GUARDED_FUNCS))
        if n_kern != len(guarded):
            sys.stderr.write(f"check_ring_kernels: {len(guarded)} guarded kernels in the log but {n_kern} in the assembly\n")
            sys.exit(1)
        # every function of the file: a packed-FP32 write of a wide buffer store's data registers right behind the store
        # (a gfx950 hazard hipcc does not pad; verify_ring_isa.store_pk_hazards)
        haz = vri.store_pk_hazards(open(sys.argv[2], errors="replace").read())
        for func, line, st, follower in haz:
            sys.stderr.write(f"wide store followed by a packed write of its data: {func}\n    .s line {line}: {st}\n        {follower}\n")
        n_isa += len(haz)
        print(f"ring kernels: {len(guarded)} guarded, {sum(r['asm_loads'] for r in res.values())} asm loads verified, "
              f"{sum(len(r['assumptions']) for r in res.values())} store-skip branches assumed not taken, "
              f"{len(bad)} spilling, {n_isa} in-flight register accesses / store hazards")
    sys.exit(1 if (bad or n_isa) else 0)
