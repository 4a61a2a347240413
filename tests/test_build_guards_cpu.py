"""The build-time guards that need no GPU: the wide-store / packed-write hazard scan (csrc/verify_ring_isa.py
store_pk_hazards: a gfx950 hazard hipcc does not pad, measured in tools/micro/store_hazard2.hip) on hand-written
assembly, and the generated (4, 2) Riccati step (csrc/gen_tile16m2_step.py): the committed tile16m2_step_gen.inc is what
the generator prints, every instruction of it keeps its distance from its producers, and no statement exceeds the
compiler's 30 asm operands."""
import importlib.util
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "iterative-linear-quadratic-regulator_amd", "csrc")


def _vri():
    spec = importlib.util.spec_from_file_location("ilqr_verify_ring_isa_test", os.path.join(CSRC, "verify_ring_isa.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_store_hazard_scan_counts_wait_states():
    v = _vri()
    asm = """
_Zfoo:
	buffer_store_dwordx4 v[240:243], v168, s[4:7], s29 offen
	v_pk_fma_f32 v[240:241], v[100:101], v[134:135], v[240:241]
	global_store_dwordx4 v[2:3], v[10:13], off
	s_nop 0
	v_pk_add_f32 v[12:13], v[1:2], v[3:4]
	global_store_dwordx4 v[2:3], v[10:13], off
	s_nop 1
	v_pk_add_f32 v[12:13], v[1:2], v[3:4]
	buffer_store_dwordx4 v[20:23], v168, s[4:7], 0 offen
	v_add_f32 v1, v2, v3
	v_pk_add_f32 v[20:21], v[1:2], v[3:4]
	buffer_store_dwordx4 v[20:23], v168, s[4:7], s3 offen
	v_add_f32 v1, v2, v3
	v_pk_add_f32 v[20:21], v[1:2], v[3:4]
	buffer_store_dwordx4 v[30:33], v168, s[4:7], s3 offen
	v_pk_add_f32 v[40:41], v[1:2], v[3:4]
	v_fma_f64 v[30:31], v[1:2], v[3:4], v[5:6]
	buffer_store_dwordx2 v[50:51], v168, s[4:7], s3 offen
	v_pk_add_f32 v[50:51], v[1:2], v[3:4]
_Zbar:
	buffer_store_dwordx4 v[60:63], v168, s[4:7], s3 offen
	s_cbranch_scc1 .LBB1_2
	v_pk_add_f32 v[60:61], v[1:2], v[3:4]
"""
    hits = v.store_pk_hazards(asm)
    # SGPR soffset + packed write directly behind; global store with one wait state; literal soffset with one instruction between
    assert [(h[0], h[1]) for h in hits] == [("_Zfoo", 3), ("_Zfoo", 5), ("_Zfoo", 11)], hits


def test_generated_42_step_is_current_and_well_formed():
    gen = os.path.join(CSRC, "gen_tile16m2_step.py")
    r = subprocess.run([sys.executable, gen], capture_output=True, text=True, cwd=CSRC)
    assert r.returncode == 0, r.stderr
    committed = open(os.path.join(CSRC, "tile16m2_step_gen.inc")).read()
    assert r.stdout == committed, "tile16m2_step_gen.inc is stale: python3 gen_tile16m2_step.py > tile16m2_step_gen.inc"
    m = re.search(r"(\d+) instructions in (\d+) issue slots \((\d+) forced wait states, (\d+) slot", committed)
    assert m and int(m.group(1)) == int(m.group(2)) and int(m.group(3)) == 0 and int(m.group(4)) == 0
    # at most 30 operands per asm statement (the compiler's limit), every statement volatile
    for stmt in committed.split("asm volatile(")[1:]:
        head = stmt.split(");")[0]
        assert len(re.findall(r'\[\w+\] "[=+&vs]+"\(', head)) <= 30
    # every DPP read of a value produced inside the stream is at least 3 instructions behind its producer (the two wait
    # states the hardware demands; nothing inside an asm statement is padded)
    lines = [l.strip().strip('"').replace("\\n\\t", "") for l in committed.splitlines() if l.strip().startswith('"v_') or l.strip().startswith('"s_nop')]
    last_write = {}
    for k, l in enumerate(lines):
        ops = re.findall(r"%\[(\w+)\]", l)
        if not ops:
            continue
        if "_dpp" in l.split()[0]:
            src0 = ops[1]                      # the operand moved across lanes is the first source
            if src0 in last_write:
                assert k - last_write[src0] >= 3, (k, l)
        last_write[ops[0]] = k
