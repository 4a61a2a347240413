"""The hardware fact behind the build's wide-store guard (csrc/verify_ring_isa.py store_pk_hazards, DESIGN 4 "a hardware
hazard hipcc does not pad"), measured on the card the tests run on: tools/micro/store_hazard2.hip issues a 16-byte store
and a packed-FP32 write of its data registers back to back inside one asm statement, with 0..4 wait states in between,
and counts stored dwords that came out as the NEW value.  The assertions are the table the guard's rule is built on:
an SGPR-offset buffer store needs one wait state before a packed write (hipcc pads none), a literal-offset buffer store
and a global store need two (hipcc pads one); fp64 writes need what hipcc pads."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wide_store_then_packed_write_needs_one_more_wait_state():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    d = tempfile.mkdtemp(prefix="store_hazard.")
    exe = os.path.join(d, "store_hazard2")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-o", exe, os.path.join(ROOT, "tools", "micro", "store_hazard2.hip")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout
    shutil.rmtree(d, ignore_errors=True)
    rows = {}
    for line in out.splitlines():
        m = re.match(r"(.+?)\s*\|\s*(\S.*?)\s*\|\s*clobbered\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+) of (\d+)", line)
        if m:
            rows[(m.group(1).strip(), m.group(2).strip())] = [int(m.group(k)) for k in (3, 4, 5, 6)]
    assert len(rows) >= 60, out[-2000:]
    sg, lit, glb = "buffer x4, SGPR soffset", "buffer x4, literal soffset", "global_store_dwordx4"
    # packed write right behind an SGPR-offset store: the high dword of the pair is the new value in part of the lanes
    assert rows[(sg, "nothing")][1] > 0 and rows[(sg, "nothing")][0] == 0
    for gap in ("s_nop 0", "s_nop 1", "1 VALU", "2 VALU"):
        assert sum(rows[(sg, gap)]) == 0, (gap, rows[(sg, gap)])
    # literal offset / global store: one wait state (what hipcc pads) is not enough, two are
    for kind in (lit, glb):
        assert rows[(kind, "nothing")][1] > 0 and rows[(kind, "s_nop 0")][1] > 0 and rows[(kind, "1 VALU")][1] > 0
        for gap in ("s_nop 1", "s_nop 2", "2 VALU"):
            assert sum(rows[(kind, gap)]) == 0, (kind, gap, rows[(kind, gap)])
    # fp64 followers: what hipcc pads is enough
    for kind in ("f64 fma/add: buffer SGPR", ):
        assert sum(rows[(kind, "nothing")]) == 0
    for kind in ("f64 fma/add: buffer lit", "f64 fma/add: global"):
        assert sum(rows[(kind, "s_nop 0")]) == 0 and sum(rows[(kind, "nothing")]) > 0
