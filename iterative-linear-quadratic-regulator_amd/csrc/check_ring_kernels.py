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

usage: check_ring_kernels.py <hipcc stderr log>      (exit 1 and a list on violation)
"""
import re
import sys

GUARDED = ("forward_ring_kernel", "backward_tile16_kernel", "backward_tile16m2_kernel")


def parse(log_text):
    """-> list of dicts {name, vgprs, agprs, scratch, vspill} for every kernel in a resource-usage log."""
    out, cur = [], None
    for line in log_text.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1), "vgprs": 0, "agprs": 0, "scratch": 0, "vspill": 0}
            out.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("vgprs", r"remark:\s+VGPRs: (\d+)"), ("agprs", r"remark:\s+AGPRs: (\d+)"),
                         ("scratch", r"remark:\s+ScratchSize \[bytes/lane\]: (\d+)"),
                         ("vspill", r"remark:\s+VGPRs Spill: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    return out


def violations(kernels):
    return [k for k in kernels if any(g in k["name"] for g in GUARDED) and (k["agprs"] or k["vspill"])]


def forward_ring_integrator(name):
    """Integrator template argument of a mangled forward_ring_kernel<T, Dyn, INTEG> name, or None."""
    if "forward_ring_kernel" not in name:
        return None
    m = re.search(r"ELi(\d+)EEEvNS_5KArgs", name)
    return int(m.group(1)) if m else None


if __name__ == "__main__":
    ks = parse(open(sys.argv[1], errors="replace").read())
    guarded = [k for k in ks if any(g in k["name"] for g in GUARDED)]
    bad = violations(ks)
    for k in bad:
        sys.stderr.write(f"ring kernel spills (agprs {k['agprs']}, vgpr spills {k['vspill']}): {k['name']}\n")
    if not guarded:
        sys.stderr.write("check_ring_kernels: no guarded kernel found in the log (was the remark flag passed?)\n")
        sys.exit(1)
    sys.exit(1 if bad else 0)
