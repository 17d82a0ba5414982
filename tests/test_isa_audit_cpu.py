"""The hand-placed LDS reads of kernels_mfma16.h, checked in the compiled gfx950 ISA (no GPU needed: hipcc cross-compiles).

hipcc does not count an `asm volatile` ds_read_b128: nothing may read its destination before the explicit
`s_waitcnt lgkmcnt(N)` that covers it - not an MFMA (stale fragment) and not a register copy the compiler makes where
control flow merges (that very bug gave exact answers only through the fall-back scan while the steady tile loop was
being written).  tools/audit_ring.py walks every mfma16_topk_kernel instantiation of the device assembly."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "theoremsearch_amd", "csrc")
ASM_DIR = os.path.join(CSRC, "build", "asm")
UNITS = ("launch_mfma16", "launch_mfma16_f32")        # the translation units that instantiate mfma16_topk_kernel


@pytest.mark.timeout(900)
def test_no_register_of_the_fragment_ring_is_read_while_its_load_is_in_flight():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which("hipcc")):
        pytest.skip("hipcc not available")
    r = subprocess.run(["make", "-C", CSRC, "asm"], capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import audit_ring
        for unit in UNITS:
            assert audit_ring.main(os.path.join(ASM_DIR, f"{unit}-hip-amdgcn-amd-amdhsa-gfx950.s")) == 0, unit
        # the register budget the kernel is written for: no scratch, one wave per SIMD
        usage = open(os.path.join(ASM_DIR, "launch_mfma16.resource_usage.txt")).read()
        blocks = usage.split("Function Name: ")
        mine = [b for b in blocks if b.startswith("_ZN2ts18mfma16_topk_kernelILi768ELi4ELi0ELb0ELb0ELb0EEE")]
        assert mine, "headline instantiation not found in the resource report"
        assert "ScratchSize [bytes/lane]: 0" in mine[0] and "VGPRs Spill: 0" in mine[0]
        pair = [b for b in blocks if b.startswith("_ZN2ts18mfma16_topk_kernelILi1024ELi2ELi0ELb0ELb0ELb1EEE")]     # the paired pass of d = 1024
        assert pair and "ScratchSize [bytes/lane]: 0" in pair[0]
    finally:
        shutil.rmtree(ASM_DIR, ignore_errors=True)   # tens of MB of intermediates: not left in the tree
