"""CPU checks of k_fused_wide64's build (no GPU): the generated code must not touch the destination of an asm LDS read
before the wait that names it (tools/asm_audit.py - the compiler sees such a register as written at once), must not
spill, and the kernel ids / names stay in step with include/evc.h."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_generated_code_passes_the_asm_audit():
    import asm_audit
    assert asm_audit.main() == 0


def test_kernel_ids_and_names():
    from exemplars_vc_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "evc.h")).read()
    ids = dict((n, int(v)) for n, v in re.findall(r"EVC_KERNEL_(\w+)\s*=\s*(\d+)", hdr))
    assert ids["FUSED_WIDE64"] == 7 and _lib.KERNEL_NAMES[7] == "k_fused_wide64"
    assert sorted(ids.values()) == sorted(_lib.KERNEL_NAMES.keys())


def test_the_audit_flags_a_touched_pending_register_and_scratch():
    """the auditor itself: a copy of a pending asm ds_read destination, and a kernel with scratch, must be reported"""
    import asm_audit
    good = """
_ZN3evc14k_fused_wide64ILi8EEEvNS_10Wide64ArgsE:
	;;#ASMSTART
	ds_read_b128 v[10:13], v5
	;;#ASMEND
	v_mfma_f64_16x16x4_f64 v[20:27], v[30:31], a[0:1], v[20:27]
	;;#ASMSTART
	s_waitcnt lgkmcnt(0)
	;;#ASMEND
	v_mov_b32_e32 v40, v10
	.amdhsa_private_segment_fixed_size 0
"""
    assert asm_audit.audit(good) == []
    bad = good.replace("v_mfma_f64_16x16x4_f64 v[20:27], v[30:31], a[0:1], v[20:27]", "v_accvgpr_write_b32 a7, v12")
    assert len(asm_audit.audit(bad)) == 1
    assert any("scratch" in b for b in asm_audit.audit(good.replace("fixed_size 0", "fixed_size 24")))
