"""The build-time guard of the kernels that count their own memory operations (csrc/check_ring_kernels.py):
its log parser, its verdicts, and -- when the library was built here -- the verdict on the real build logs."""
import importlib.util
import os

import pytest

from ilqr_amd import _lib

spec = importlib.util.spec_from_file_location("crk", os.path.join(_lib.CSRC, "check_ring_kernels.py"))
crk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(crk)

LOG = """
k.hpp:1:1: remark: Function Name: _ZN4ilqr19forward_ring_kernelIfNS_8PendulumIfEELi2EEEvNS_5KArgsIT_EE [-Rpass-analysis=kernel-resource-usage]
k.hpp:1:1: remark:     VGPRs: 120 [-Rpass-analysis=kernel-resource-usage]
k.hpp:1:1: remark:     AGPRs: 0 [-Rpass-analysis=kernel-resource-usage]
k.hpp:1:1: remark:     ScratchSize [bytes/lane]: 20 [-Rpass-analysis=kernel-resource-usage]
k.hpp:1:1: remark:     VGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]
k.hpp:1:1: remark: Function Name: _ZN4ilqr19forward_ring_kernelIdNS_7UserDynIdEELi3EEEvNS_5KArgsIT_EE [-Rpass-analysis=kernel-resource-usage]
k.hpp:1:1: remark:     VGPRs: 256 [-Rpass-analysis=kernel-resource-usage]
k.hpp:1:1: remark:     AGPRs: 60 [-Rpass-analysis=kernel-resource-usage]
k.hpp:1:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass-analysis=kernel-resource-usage]
k.hpp:1:1: remark:     VGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]
k.hpp:1:1: remark: Function Name: _ZN4ilqr13select_kernelIfEEvNS_5KArgsIT_EE [-Rpass-analysis=kernel-resource-usage]
k.hpp:1:1: remark:     VGPRs: 20 [-Rpass-analysis=kernel-resource-usage]
k.hpp:1:1: remark:     AGPRs: 4 [-Rpass-analysis=kernel-resource-usage]
"""


def test_parser_and_verdicts():
    ks = crk.parse(LOG)
    assert [k["vgprs"] for k in ks] == [120, 256, 20]
    bad = crk.violations(ks)
    # SGPR spills through scratch are harmless; AGPR parking of a ring kernel is not; unguarded kernels are ignored
    assert [k["name"] for k in bad] == ["_ZN4ilqr19forward_ring_kernelIdNS_7UserDynIdEELi3EEEvNS_5KArgsIT_EE"]
    assert crk.forward_ring_integrator(bad[0]["name"]) == 3
    assert crk.forward_ring_integrator(ks[2]["name"]) is None


@pytest.mark.parametrize("log", ["ilqr_f32.usage.log", "ilqr_f64.usage.log"])
def test_library_build_logs_are_clean(log):
    path = os.path.join(_lib.CSRC, log)
    if not os.path.exists(path):
        pytest.skip("library not built in this tree")
    ks = crk.parse(open(path, errors="replace").read())
    guarded = [k for k in ks if any(g in k["name"] for g in crk.GUARDED)]
    assert len(guarded) >= 20 and not crk.violations(ks)


# ---- the ISA-level verifier (csrc/verify_ring_isa.py): vmcnt queue simulation on the compiler's assembly -----------
spec2 = importlib.util.spec_from_file_location("vri", os.path.join(_lib.CSRC, "verify_ring_isa.py"))
vri = importlib.util.module_from_spec(spec2)
spec2.loader.exec_module(vri)

_HEAD = """
_ZN4ilqr22backward_tile16_kernelIfLb0ELi4EEEvNS_5KArgsIT_EE:
	s_load_dwordx2 s[0:1], s[4:5], 0x0
	;;#ASMSTART
	s_nop 4
	buffer_load_dwordx4 v[10:13], v1, s[8:11], s2 offen
	buffer_load_dword v20, v1, s[8:11], s2 offen offset:64
	;;#ASMEND
	;;#ASMSTART
	s_nop 4
	buffer_load_dwordx4 v[14:17], v1, s[8:11], s3 offen
	buffer_load_dword v21, v1, s[8:11], s3 offen offset:64
	;;#ASMEND
"""
_TAIL = """
	s_endpgm
.Lfunc_end0:
"""


def _verdict(body):
    res = vri.verify_text(_HEAD + body + _TAIL)
    assert len(res) == 1
    return list(res.values())[0]["violations"]


def test_isa_verifier_accepts_a_counted_wait_and_rejects_early_access():
    # slot 0 (v10-13, v20) has landed once at most 2 younger operations are outstanding
    ok = """
	;;#ASMSTART
	s_waitcnt vmcnt(2)
	;;#ASMEND
	v_fma_f32 v30, v10, v11, v20
"""
    assert _verdict(ok) == []
    # ... but slot 1 (v14-17, v21) has not: a plain register copy of it right behind the issue is the round-1 fp64 bug
    copy = ok + "	v_mov_b32_e32 v40, v21\n"
    v = _verdict(copy)
    assert len(v) == 1 and v[0][2] == [21] and "v_mov_b32_e32" in v[0][1]
    # one wait state too few is caught as well
    assert _verdict(ok.replace("vmcnt(2)", "vmcnt(3)")) != []
    # a store issued in between moves the count (loads and stores retire in issue order)
    stored = """
	buffer_store_dword v50, v2, s[12:15], s6 offen
	;;#ASMSTART
	s_waitcnt vmcnt(3)
	;;#ASMEND
	v_fma_f32 v30, v10, v11, v20
"""
    assert _verdict(stored) == []


def test_isa_verifier_follows_loops_and_both_sides_of_branches():
    # the refill of slot 0 inside a loop is consumed one trip later: the back edge must carry the queue
    loop = """
.LBB0_1:
	;;#ASMSTART
	s_waitcnt vmcnt(2)
	;;#ASMEND
	v_fma_f32 v30, v10, v11, v20
	;;#ASMSTART
	s_nop 4
	buffer_load_dwordx4 v[10:13], v1, s[8:11], s2 offen
	buffer_load_dword v20, v1, s[8:11], s2 offen offset:64
	;;#ASMEND
	;;#ASMSTART
	s_waitcnt vmcnt(2)
	;;#ASMEND
	v_fma_f32 v31, v14, v15, v21
	;;#ASMSTART
	s_nop 4
	buffer_load_dwordx4 v[14:17], v1, s[8:11], s3 offen
	buffer_load_dword v21, v1, s[8:11], s3 offen offset:64
	;;#ASMEND
	s_cbranch_scc1 .LBB0_1
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
"""
    assert _verdict(loop) == []
    assert _verdict(loop.replace("v_fma_f32 v31, v14, v15, v21", "v_fma_f32 v31, v14, v15, v20")) != []
    # packed math that broadcasts ONE half of a register pair reads only that register
    pk = """
	;;#ASMSTART
	s_waitcnt vmcnt(2)
	;;#ASMEND
	v_pk_fma_f32 v[30:31], v[20:21], v[32:33], v[34:35] op_sel_hi:[0,1,1]
"""
    assert _verdict(pk) == []
    assert _verdict(pk.replace("op_sel_hi:[0,1,1]", "op_sel_hi:[1,1,1]")) != []
    # v_pk_mov_b32 reads one half of each source: D.lo = src0[op_sel[0]], D.hi = src1[op_sel[1]]
    mov = pk.replace("v_pk_fma_f32 v[30:31], v[20:21], v[32:33], v[34:35] op_sel_hi:[0,1,1]",
                     "v_pk_mov_b32 v[30:31], v[32:33], v[20:21] op_sel:[1,0]")
    assert _verdict(mov) == []                                   # reads v33 and v20; v21 is the load in flight
    assert _verdict(mov.replace("op_sel:[1,0]", "op_sel:[1,1]")) != []


@pytest.mark.parametrize("log", ["ilqr_f32.isa.log", "ilqr_f64.isa.log"])
def test_library_assembly_was_verified(log):
    path = os.path.join(_lib.CSRC, log)
    if not os.path.exists(path):
        pytest.skip("library not built in this tree")
    line = open(path).read().strip().splitlines()[-1]
    assert "0 spilling, 0 in-flight register accesses" in line and "asm loads verified" in line, line
