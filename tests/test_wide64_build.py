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
