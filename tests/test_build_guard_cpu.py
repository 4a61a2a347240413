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
