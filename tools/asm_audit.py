"""Audit of k_fused_wide64's generated code: between an asm ds_read (its destination counts as written at ;;#ASMEND for
the compiler) and the asm s_waitcnt lgkmcnt(0) that follows it, no instruction may read or write the destination
registers - a copy or a reuse there would take stale data (cdna_hip_programming.md, 'Inline asm').
    python tools/asm_audit.py        (compiles evc_wide64.hip with -save-temps into a temporary directory)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "exemplars_vc_amd", "csrc", "evc_wide64.hip")


def audit(asm_text):
    bad = []
    for m in re.finditer(r"^(_ZN3evc14k_fused_wide64ILi\d+E\S*):(.*?)\.amdhsa_private_segment_fixed_size (\d+)", asm_text, re.S | re.M):
        name, body, scratch = m.group(1), m.group(2).split("\n"), int(m.group(3))
        if scratch:
            bad.append(f"{name}: {scratch} bytes of scratch")
        pend = {}
        for n, line in enumerate(body):
            t = line.strip()
            if not t or t.startswith(";") or t.startswith("."):
                continue
            in_asm = ";;#ASMSTART" in body[n - 1]
            mm = re.match(r"ds_read_b128 v\[(\d+):(\d+)\]", t)
            if mm and in_asm:
                pend[(int(mm.group(1)), int(mm.group(2)))] = n
                continue
            if in_asm and t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                pend.clear()
                continue
            for (a, b), ln in pend.items():
                for r in re.findall(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", t):
                    lo = int(r[0]) if r[0] else int(r[2])
                    hi = int(r[1]) if r[1] else lo
                    if not (hi < a or lo > b):
                        bad.append(f"{name}: line {n}: '{t}' touches v[{a}:{b}] pending since line {ln}")
    return bad


def main():
    with tempfile.TemporaryDirectory() as d:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w", "-save-temps=obj", "-c", SRC,
               "-o", os.path.join(d, "w64.o")]
        subprocess.run(cmd, check=True, cwd=d)
        asm = [f for f in os.listdir(d) if f.endswith("gfx950.s")]
        bad = audit(open(os.path.join(d, asm[0])).read())
    for b in bad:
        print(b)
    print("asm audit:", "FAILED" if bad else "ok")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
