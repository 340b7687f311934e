#!/usr/bin/env python3
"""Timeline of the last call in a rocprofv3 kernel trace: kernel durations and the idle gaps between them.
usage: trace_gaps.py <p_kernel_trace.csv> [first kernel substring of the call]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2] if len(sys.argv) > 2 else "k_utt_setup"
starts = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
i0 = starts[-1]
t_prev = None
tot_k = 0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if t_prev is None else s - t_prev
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    print(f"{gap / 1e3:8.1f} us gap  {(e - s) / 1e3:9.1f} us  {name}")
    tot_k += e - s
    t_prev = e
print(f"kernels {tot_k / 1e3:.1f} us, span {(int(rows[-1]['End_Timestamp']) - int(rows[i0]['Start_Timestamp'])) / 1e3:.1f} us")
