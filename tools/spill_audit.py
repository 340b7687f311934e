#!/usr/bin/env python3
"""Which scratch (spill) accesses of a kernel stand inside its MFMA loops?  (CPU tool, reads hipcc -S output)

    python tools/spill_audit.py /tmp/isa/fused_xy.s _ZN3evc10k_fused_xyILi7ELb0EEEvNS_9FusedArgsE

Prints the MFMA clusters (a unit = one cluster) and every scratch access that lies between the first and the last
cluster of a run of clusters less than `gap` lines apart, i.e. inside an unrolled sweep."""
import re
import sys


def audit(path, kernel, gap=450):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    mf = [i for i, l in enumerate(body) if "v_mfma" in l]
    sc = [i for i, l in enumerate(body) if "scratch_" in l]
    clusters = []
    for i in mf:
        if clusters and i - clusters[-1][1] <= 60:
            clusters[-1][1] = i
            clusters[-1][2] += 1
        else:
            clusters.append([i, i, 1])
    runs = []
    for c in clusters:
        if runs and c[0] - runs[-1][-1][1] <= gap:
            runs[-1].append(c)
        else:
            runs.append([c])
    total = 0
    for r in runs:
        lo, hi = r[0][0], r[-1][1]
        inside = [i for i in sc if lo <= i <= hi]
        total += len(inside) if len(r) > 2 else 0
        print(f"run lines {lo}-{hi}: {len(r)} clusters, MFMAs {[c[2] for c in r]}, scratch accesses inside: {len(inside)}")
        for i in inside[:12]:
            print("     ", i, body[i].strip()[:90])
    m = re.search(r"; ScratchSize: (\d+)", "\n".join(lines[end:end + 80]))
    print("scratch bytes per lane:", m.group(1) if m else "?", " accesses inside sweeps:", total)
    return total


if __name__ == "__main__":
    audit(sys.argv[1], sys.argv[2])
