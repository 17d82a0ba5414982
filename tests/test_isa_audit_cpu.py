"""The hand-placed LDS reads of kernels_mfma16.h, checked in the compiled gfx950 ISA (no GPU needed: hipcc cross-compiles).

hipcc does not count an `asm volatile` ds_read_b128: nothing may read its destination before the explicit
`s_waitcnt lgkmcnt(N)` that covers it - not an MFMA (stale fragment) and not a register copy the compiler makes where
control flow merges (that very bug gave exact answers only through the fall-back scan while the steady tile loop was
being written).  Nor may a vector instruction of the compiler's write an MFMA operand right in front of that MFMA: the MFMAs
are asm text too, the hazard recognizer pads nothing for them (the first k-split build parked two query fragments in AGPRs and
brought them back one instruction ahead of their MFMA: those k-steps came out wrong on the GPU).  tools/audit_ring.py walks
every mfma16_topk_kernel instantiation of the device assembly for both."""
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
        mine = [b for b in blocks if b.startswith("_ZN2ts18mfma16_topk_kernelILi768ELi4ELi0ELb0ELb0ELb0ELb0EEE")]
        assert mine, "headline instantiation not found in the resource report"
        assert "ScratchSize [bytes/lane]: 0" in mine[0] and "VGPRs Spill: 0" in mine[0]
        for form in ("ILi1024ELi2ELi0ELb0ELb0ELb1ELb0EEE", "ILi1024ELi4ELi0ELb0ELb0ELb1ELb1EEE"):      # the paired pass of d = 1024: both forms
            pair = [b for b in blocks if b.startswith("_ZN2ts18mfma16_topk_kernel" + form)]
            assert pair and "ScratchSize [bytes/lane]: 0" in pair[0], form
    finally:
        shutil.rmtree(ASM_DIR, ignore_errors=True)   # tens of MB of intermediates: not left in the tree


def test_the_audit_sees_an_operand_written_right_in_front_of_its_mfma():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import audit_ring
    good = ["s_waitcnt lgkmcnt(0)", "v_accvgpr_read_b32 v174, a148", "s_nop 1", "s_nop 0",
            "v_mfma_f32_16x16x32_bf16 v[138:141], v[166:169], v[174:177], v[138:141]"]
    bad = ["s_waitcnt lgkmcnt(0)", "v_accvgpr_read_b32 v174, a148",
           "v_mfma_f32_16x16x32_bf16 v[138:141], v[166:169], v[174:177], v[138:141]"]
    other = ["v_mov_b32_e32 v9, v3", "v_mfma_f32_16x16x32_bf16 v[138:141], v[166:169], v[174:177], v[138:141]"]
    assert audit_ring.audit("k", list(enumerate(good))) == []
    assert audit_ring.audit("k", list(enumerate(other))) == []
    found = audit_ring.audit("k", list(enumerate(bad)))
    assert len(found) == 1 and "right in front of its MFMA" in found[0][1]
    # and the original check: a fragment read before its wait
    stale = ["ds_read_b128 v[10:13], v2", "v_mfma_f32_16x16x32_bf16 v[138:141], v[10:13], v[174:177], v[138:141]"]
    assert len(audit_ring.audit("k", list(enumerate(stale)))) == 1
