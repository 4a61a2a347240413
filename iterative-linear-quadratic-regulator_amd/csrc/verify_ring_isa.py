"""ISA-level verifier for the kernels that count their own memory operations.

forward_ring_kernel, backward_tile16_kernel and backward_tile16m2_kernel keep a ring of inline-asm buffer loads
in flight and wait for them with hand-counted ``s_waitcnt vmcnt(N)``.  hipcc treats the destination registers of
such a load as written at ``;;#ASMEND``: nothing in the language stops it from COPYING one of them (a ``v_mov``
that the register allocator inserts for a loop-carried value), spilling it, or reusing it before the data has
landed.  That is what broke the fp64 sweep with branch-free "dropped" stores in round 1 (wrong gains, NaN): hipcc
parked every slot's ``l_xx`` load in ONE temporary register and copied it to the slot's own register right behind
the load's ``;;#ASMEND`` -- the copy read a register whose load was still in flight.  The resource-usage guard
(check_ring_kernels.py) sees spills, not copies, so the invariant is verified here on the compiler's own assembly:

    walking every path through a guarded kernel, with the vmcnt queue simulated (loads, stores and LDS-DMA retire in
    issue order; ``s_waitcnt vmcnt(N)`` retires all but the N youngest), NO instruction may read or write a register
    that is the destination of an inline-asm load still in the queue.

The walk explores both sides of every conditional branch and memoises (pc, queue), so loops are followed until
their state repeats.  One assumption is made and reported: an ``s_cbranch_execz`` that only skips stores is
taken as not taken -- the kernels' counts assume one store per step, and every wave that reaches the sweep has a
storing lane (waves without an active trajectory leave at the top of the kernel).

usage: verify_ring_isa.py <device .s file> [kernel-name-substring ...]      exit 1 and a report on a violation
"""
import re
import sys

GUARDED = ("forward_ring_kernel", "backward_tile16_kernel", "backward_tile16m2_kernel", "forward_mfma16_kernel")

_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
_VMCNT = re.compile(r"vmcnt\((\d+)\)")
_LABEL = re.compile(r"^([.\w$]+):")
_VMEM_LOAD = ("buffer_load", "global_load", "flat_load", "scratch_load", "tbuffer_load")
_VMEM_STORE = ("buffer_store", "global_store", "flat_store", "scratch_store", "tbuffer_store")
_VMEM_ATOMIC = ("buffer_atomic", "global_atomic", "flat_atomic")


def _regs(text):
    out = set()
    for m in _VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


_OPSEL = re.compile(r"\bop_sel:\[([01,]+)\]")
_OPSELHI = re.compile(r"\bop_sel_hi:\[([01,]+)\]")


def _pk_touch(ops, mov=False):
    """Registers a packed-math instruction really accesses: a 64-bit SOURCE operand whose op_sel / op_sel_hi bits
    both select the same half reads only that register (hipcc broadcasts a scalar over both lanes this way).
    v_pk_mov_b32 (mov=True) reads ONE half of each source: D.lo = src0[op_sel[0]], D.hi = src1[op_sel[1]]."""
    body = ops.split(" op_sel")[0]
    fields = [f.strip() for f in body.split(",")]
    sel = [int(c) for c in _OPSEL.search(ops).group(1).split(",")] if _OPSEL.search(ops) else []
    hi = [int(c) for c in _OPSELHI.search(ops).group(1).split(",")] if _OPSELHI.search(ops) else []
    out = set(_regs(fields[0])) if fields else set()
    for k, f in enumerate(fields[1:]):
        r = sorted(_regs(f))
        if len(r) == 2:
            lo_sel = sel[k] if k < len(sel) else 0
            hi_sel = lo_sel if mov else (hi[k] if k < len(hi) else 1)
            if lo_sel == hi_sel:
                r = [r[lo_sel]]
        out.update(r)
    return out


class Inst:
    __slots__ = ("text", "mnem", "asm", "kind", "dest", "touch", "vmcnt", "target", "line")

    def __init__(self, text, asm, line):
        self.text, self.asm, self.line = text, asm, line
        self.mnem = text.split()[0]
        ops = text[len(self.mnem):]
        self.kind, self.dest, self.vmcnt, self.target = "other", frozenset(), None, None
        self.touch = frozenset(_regs(ops))
        m = self.mnem
        if m.startswith("v_pk_"):
            self.touch = frozenset(_pk_touch(ops, mov=(m == "v_pk_mov_b32")))
        if m.startswith(_VMEM_LOAD):
            self.kind = "load"
            if " lds" in ops or m.startswith("global_load_lds"):
                self.dest = frozenset()          # LDS-DMA: counted by vmcnt, no VGPR destination
            else:
                self.dest = frozenset(_regs(ops.split(",")[0]))
        elif m.startswith(_VMEM_STORE):
            self.kind = "store"
        elif m.startswith(_VMEM_ATOMIC):
            self.kind = "store"                  # counted like a store; a returning atomic is not used by these kernels
        elif m == "s_waitcnt":
            w = _VMCNT.search(ops)
            if w:
                self.kind, self.vmcnt = "wait", int(w.group(1))
        elif m.startswith("s_cbranch") or m == "s_branch":
            self.kind = "branch"
            self.target = ops.strip().split()[0]
        elif m in ("s_endpgm", "s_endpgm_saved", "s_setpc_b64"):
            self.kind = "end"                    # (s_setpc_b64: the return of a non-kernel function, e.g. role_rollout)


def parse_kernels(asm_text, names=GUARDED):
    """-> {kernel symbol: (list of Inst, {label: index})} for every kernel whose symbol contains one of `names`."""
    kernels, cur, labels, in_asm, name = {}, None, None, False, None
    for ln, raw in enumerate(asm_text.splitlines(), 1):
        line = raw.strip()
        if cur is None:
            m = _LABEL.match(line)
            if m and m.group(1).startswith("_Z") and any(n in m.group(1) for n in names):
                name, cur, labels, in_asm = m.group(1), [], {}, False
            continue
        if line.startswith(".Lfunc_end"):
            kernels[name] = (cur, labels)
            cur = None
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line or line.startswith(";") or line.startswith("//"):
            continue
        m = _LABEL.match(line)
        if m:
            labels[m.group(1)] = len(cur)
            continue
        if line.startswith("."):
            continue
        text = line.split(";")[0].strip()
        if text:
            cur.append(Inst(text, in_asm, ln))
    return kernels


def _skips_only_stores(insts, labels, i):
    """insts[i] is an s_cbranch_execz: does its not-taken side hold stores and nothing else that vmcnt counts, and
    then join the taken side?  (Either the skipped block lies in front of the target, or hipcc moved it out of line
    and it ends with an s_branch to the same target.)"""
    tgt = labels.get(insts[i].target)
    stores, j = 0, i + 1
    while j < len(insts):
        x = insts[j]
        if j == tgt:
            return stores > 0
        if x.kind == "branch":
            return stores > 0 and x.mnem == "s_branch" and labels.get(x.target) == tgt
        # (a store written as inline asm with its wait state -- buf_store_vec -- is a store like any other)
        if x.kind == "load" or (x.asm and x.kind != "store" and x.mnem != "s_nop") or x.kind in ("end", "wait"):
            return False
        stores += x.kind == "store"
        j += 1
    return False


def verify_kernel(insts, labels, max_states=400000):
    """-> (violations, assumptions).  A violation = (line, text, in-flight registers touched)."""
    violations, assumptions, seen = {}, set(), set()
    stack = [(0, ())]          # (pc, queue); queue entry = ("A", dest regs of an asm load) | ("c",) anything else
    while stack:
        pc, q = stack.pop()
        while True:
            if pc >= len(insts):
                break
            key = (pc, q)
            if key in seen:
                break
            seen.add(key)
            if len(seen) > max_states:
                raise RuntimeError("verify_ring_isa: state space too large")
            ins = insts[pc]
            inflight = frozenset().union(*[e[1] for e in q if e[0] == "A"]) if q else frozenset()
            hit = ins.touch & inflight
            if hit and ins.kind != "wait":
                violations.setdefault(ins.line, (ins.line, ins.text, sorted(hit)))
            if ins.kind == "load":
                q = q + ((("A", ins.dest) if (ins.asm and ins.dest) else ("c",)),)
            elif ins.kind == "store":
                q = q + (("c",),)
            elif ins.kind == "wait":
                if len(q) > ins.vmcnt:
                    q = q[len(q) - ins.vmcnt:] if ins.vmcnt else ()
            elif ins.kind == "end":
                break
            elif ins.kind == "branch":
                tgt = labels.get(ins.target)
                if tgt is None:
                    break                                   # a branch out of the kernel body (does not occur)
                if ins.mnem == "s_branch":
                    pc = tgt
                    continue
                if ins.mnem == "s_cbranch_execz" and _skips_only_stores(insts, labels, pc):
                    assumptions.add((ins.line, ins.text))   # see the module docstring
                else:
                    stack.append((tgt, q))
            pc += 1
    return sorted(violations.values()), sorted(assumptions)


def verify_text(asm_text, names=GUARDED):
    """-> {kernel: {"violations": [...], "assumptions": [...], "asm_loads": n}} for the guarded kernels of a .s file."""
    out = {}
    for name, (insts, labels) in parse_kernels(asm_text, names).items():
        v, a = verify_kernel(insts, labels)
        out[name] = {"violations": v, "assumptions": a,
                     "asm_loads": sum(1 for x in insts if x.kind == "load" and x.asm and x.dest)}
    return out



# ---- the wide-store / packed-write hazard (gfx950, ROCm 7.2; measured: tools/micro/store_hazard.hip, store_hazard2.hip) ----
# A vector-memory store of more than 64 bits reads its data registers over several cycles.  The documented hazard -- a
# VALU write of those registers needs one wait state behind the store unless the store has an SGPR soffset -- is what
# LLVM pads (GCNHazardRecognizer::createsVALUHazard).  On gfx950 a PACKED-FP32 instruction (v_pk_add/mul/fma_f32: two
# passes) writing v[a:a+1] or v[a+2:a+3] needs ONE MORE: measured, the stored high dword is the new value in ~25 % of the
# lanes with
#     buffer_store_dwordx4 (SGPR soffset)                   followed directly by the packed write   (hipcc pads nothing)
#     buffer_store_dwordx4 (literal soffset) / global_store_dwordx4   with one wait state in between (hipcc pads one)
# and never with one / two wait states respectively (an s_nop or any other instruction counts).  64-bit stores, LDS
# writes, unpacked and fp64 instructions are not affected.  The persistent kernel's rollout role hit it (x_1 of the first
# N % PF steps of a candidate, found by the bit-equality test against the separate kernels).  Nothing in the language
# controls what hipcc schedules behind a store, so EVERY function of the library is checked here, and the 16-byte stores
# of the library are asm statements that carry their own wait states (buf_store_vec / vec_store).
_WIDE_STORE = re.compile(r"^\s*((?:buffer|global|flat|scratch)_store_dwordx[34])\s+(.*)$")
_FUNC = re.compile(r"^([A-Za-z_][\w$.]*):")


def _wide_store(st):
    """(data registers, wait states a packed write of them needs) of a wide store instruction, or None."""
    m = _WIDE_STORE.match(st)
    if not m:
        return None
    ops = [o.strip() for o in m.group(2).split(",")]
    if m.group(1).startswith("buffer"):
        data = _regs(ops[0])
        soff = ops[3].split()[0] if len(ops) > 3 else "0"
        return data, (1 if re.match(r"^s\d+$", soff) else 2)
    return _regs(ops[1]), 2            # global / flat / scratch: address first, data second


def store_pk_hazards(asm_text):
    """-> list of (function, line number, store text, follower text) for every wide store with a packed VALU write of its
    data registers fewer wait states behind it than that store needs (every instruction in between counts one, s_nop N
    counts N + 1).  Straight-line scan; a label does not separate the pair, a branch ends the window."""
    out = []
    func, pending = None, []
    for ln, raw in enumerate(asm_text.splitlines(), 1):
        line = raw.split(";")[0].rstrip()
        st = line.strip()
        if not st:
            continue
        m = _FUNC.match(line)
        if m and not line.startswith(".L"):
            func, pending = m.group(1), []
        if st.startswith(".") or st.endswith(":"):     # directives and labels
            continue
        if st.startswith("v_pk_") and pending:
            dst = _regs(st.split(None, 1)[1].split(",")[0])
            for data, need, text, l0 in pending:
                if need > 0 and dst & data:
                    out.append((func, l0, text, st))
        gap = 1
        if st.startswith("s_nop"):
            gap = int(st.split()[1], 0) + 1
        pending = [(d, n - gap, t, l) for d, n, t, l in pending if n - gap > 0]
        if st.startswith(("s_branch", "s_cbranch", "s_setpc", "s_endpgm", "s_swappc")):
            pending = []
        w = _wide_store(st)
        if w:
            pending.append((w[0], w[1], st, ln))
    return out


def report(results, stream=sys.stderr):
    bad = 0
    for name, r in results.items():
        for line, text, regs in r["violations"]:
            bad += 1
            stream.write(f"in-flight asm-load register touched: {name}\n    .s line {line}: {text}    (v{regs})\n")
    return bad


if __name__ == "__main__":
    res = verify_text(open(sys.argv[1], errors="replace").read(), tuple(sys.argv[2:]) or GUARDED)
    if not res:
        sys.stderr.write("verify_ring_isa: no guarded kernel found in the file\n")
        sys.exit(1)
    n_bad = report(res)
    n_assume = sum(len(r["assumptions"]) for r in res.values())
    print(f"verify_ring_isa: {len(res)} guarded kernels, {sum(r['asm_loads'] for r in res.values())} asm loads, "
          f"{n_assume} store-skip branches assumed not taken, {n_bad} violations")
    sys.exit(1 if n_bad else 0)
