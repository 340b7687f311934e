#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output per kernel (evc kernels only): mean per dispatch.
usage: pmc_summary.py <pmc dir> [--json out.json M N K frames dtype kernel_tag [steps_per_pass]]
(the json is what bench.py reads for roofline.traffic: HBM bytes per launch of the dominant kernel, with the
gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md applied: FETCH_SIZE x 2 for wide coalesced reads,
WRITE_SIZE exact; both counters are reported in KiB)"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "evc::" not in k:
            continue
        k = k.split("(")[0].replace("void ", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")

if "--json" in sys.argv:
    i = sys.argv.index("--json")
    dest = sys.argv[i + 1]
    M, N, K, frames = map(int, sys.argv[i + 2:i + 6])
    dtype, tag = sys.argv[i + 6], sys.argv[i + 7]
    # the dominant kernel: the one matching `tag` that issued the most MFMAs in total
    name = max((k for k in acc if tag in k), key=lambda k: sum(acc[k].get("SQ_INSTS_MFMA", [0])))
    c = {n: sum(v) / len(v) for n, v in acc[name].items()}
    w = 8 if dtype == "f64" else 4
    if "k_fused_wide" in name:      # H and P stream once per iteration (H read + written, P read); P and H0 written once
        algorithmic = int(frames * N * w * (3 * K + 2))
        note = ("k_fused_wide: per iteration H read and written and P read once, whole launch = K iterations + the "
                "pass that forms P; dictionary blocks from L2, partial V' exchange not counted")
    elif "k_fused_all" in name:     # H read once and written once per launch; nothing else streams
        algorithmic = frames * N * w * 2
        note = "k_fused_all: activations read once and written once per launch, dictionary fragments from L2"
    elif "k_fused_res" in name:     # every other exemplar tile streams (read + write) per iteration
        algorithmic = int(frames * N * w * (2 * 0.5 * K + 2 * 0.5))
        note = "k_fused_res: half of the activation tiles stream once per iteration"
    else:                           # the update contraction: H and P read, H written, per launch (one iteration)
        algorithmic = frames * N * w * 3
        note = "update contraction: H and P read, H' written once per launch (= per iteration)"
    doc = {
        "kernel": name.replace("evc::", ""),
        "workload": {"M": M, "N": N, "K": K, "frames": frames, "dtype": dtype},
        "source": "rocprofv3 --pmc, separate passes (tools/prof_case.sh / tools/pmc_fused.sh): FETCH_SIZE; "
                  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum; SQ_*",
        "FETCH_SIZE_kb": c.get("FETCH_SIZE"), "WRITE_SIZE_kb": c.get("WRITE_SIZE"),
        "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced reads -> x2 "
                      "(MI355X_MICROARCH.md, HBM); WRITE_SIZE exact for 16 B/lane stores",
        "hbm_bytes_per_launch": 1024.0 * (2 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)),
        "algorithmic_hbm_bytes_per_launch": algorithmic, "algorithmic_note": note,
        "mfma_busy_cycles": c.get("SQ_VALU_MFMA_BUSY_CYCLES"),
        "gui_active_cycles_sum_xcd": c.get("GRBM_GUI_ACTIVE"),
        # busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs (128 SIMDs each)
        "mfma_pipe_occupancy": (c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 128.0)
                                if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE") else None),
        "valu_insts_per_mfma": (c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"]
                                if c.get("SQ_INSTS_MFMA") else None),
    }
    if "k_fused" not in name:
        # generic path: bench.py's timed "launch" is the whole iteration loop (two contractions and, for short
        # batches, a slab sum per iteration), so the traffic on record is the loop's: every kernel matching the
        # tag (plus k_sum_slabs), bytes per dispatch x dispatches per bench step
        steps = int(sys.argv[i + 8]) if len(sys.argv) > i + 8 else 2      # prof_bench.sh: --steps 1 --warmup 1
        per_kernel, total = {}, 0.0
        for k in acc:
            if not (tag in k or "k_sum_slabs" in k) or "FETCH_SIZE" not in acc[k]:
                continue
            ck = {n: sum(v) / len(v) for n, v in acc[k].items()}
            calls = len(acc[k]["FETCH_SIZE"]) / steps
            if calls < K / 2:                      # (set-up products outside the loop)
                continue
            b = 1024.0 * (2 * ck.get("FETCH_SIZE", 0.0) + ck.get("WRITE_SIZE", 0.0))
            total += b * calls
            per_kernel[k.replace("evc::", "")] = {
                "calls_per_step": calls, "hbm_bytes_per_call": b,
                "mfma_pipe_occupancy": (ck["SQ_VALU_MFMA_BUSY_CYCLES"] / (ck["GRBM_GUI_ACTIVE"] * 128.0)
                                        if ck.get("SQ_VALU_MFMA_BUSY_CYCLES") and ck.get("GRBM_GUI_ACTIVE") else None),
                "valu_insts_per_mfma": (ck["SQ_INSTS_VALU"] / ck["SQ_INSTS_MFMA"] if ck.get("SQ_INSTS_MFMA") else None)}
        doc["kernel"] = tag + " iteration loop: " + " + ".join(sorted(per_kernel))
        doc["kernels"] = per_kernel
        doc["hbm_bytes_per_launch"] = total
        doc["algorithmic_hbm_bytes_per_launch"] = frames * N * w * 4 * K
        doc["algorithmic_note"] = ("per iteration: V = H Am^T reads H; the update contraction reads H and P and "
                                   "writes H': 4 x frames x N x element size, x K iterations = one bench launch (loop)")
        for key in ("FETCH_SIZE_kb", "WRITE_SIZE_kb", "mfma_busy_cycles", "gui_active_cycles_sum_xcd",
                    "mfma_pipe_occupancy", "valu_insts_per_mfma"):
            doc.pop(key, None)
    json.dump(doc, open(dest, "w"), indent=1)
