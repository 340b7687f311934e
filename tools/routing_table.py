#!/usr/bin/env python3
"""profiles/r04_tune_routing_k{20,80}.jsonl (tools/tune_routing.py) -> profiles/r04_routing_table.md"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def table(path, out):
    rows = [json.loads(l) for l in open(path)]
    K = rows[0]["K"]
    for dt in ("f32", "f64"):
        for M in sorted({r["M"] for r in rows if r["dtype"] == dt}):
            Ns = sorted({r["N"] for r in rows if r["dtype"] == dt and r["M"] == M})
            out.append(f"\n**{dt}, M = {M}, K = {K}** - fraction of the {'fp32' if dt == 'f32' else 'fp64'} matrix peak over the whole call, "
                       "fused / two contractions (bold: the faster)\n")
            out.append("| utterances (frame tiles) | " + " | ".join(f"N = {n}" for n in Ns) + " |")
            out.append("|---|" + "---|" * len(Ns))
            for U in sorted({r["utterances"] for r in rows}):
                cells = []
                for n in Ns:
                    rr = [r for r in rows if r["dtype"] == dt and r["M"] == M and r["N"] == n and r["utterances"] == U]
                    if not rr:
                        cells.append("-")
                        continue
                    f, g = rr[0]["fused"]["frac"], rr[0]["two_contractions"]["frac"]
                    cells.append(f"**{f:.3f}** / {g:.3f}" if f > g else f"{f:.3f} / **{g:.3f}**")
                out.append(f"| {U} ({U * 43}) | " + " | ".join(cells) + " |")


def main():
    out = ["# Fused task-queue kernels against the two contractions over N, M and the batch size (round 4, VERDICT r03 item 5)",
           "",
           "`tools/tune_routing.py` on one MI355X: whole `evc_nmf_solve` calls (import, packing, first pass, K iterations, export of",
           "`H`), device-resident inputs, best of two timed runs after a warm-up; `fused` = `k_fused_wide` (float32) / `k_fused_wide64`",
           "(float64) forced through the tuning bits, `two contractions` = `k_gemm2` / `k_gemm_nt` (`EVC_FLAG_NO_FUSED`). Utterances of",
           "688 frames.  Raw lines: `r04_tune_routing_k20.jsonl`, `r04_tune_routing_k80.jsonl`.  `use_wide` (`evc_api.hip`) is written",
           "from these tables; the fractions are of whole calls, so they sit below `bench.py`'s loop-only `roofline.frac`.",
           "",
           "Every row is of the final build of round 4 (first tickets by index, later tickets drawn when a task is finished, static",
           "schedule for small batches - float32 with tagged hand-offs -, 8 wavefronts per workgroup, `k_fused_wide64<3>`), except",
           "M = 160, which predates the late tickets.  An earlier float32 sweep of this round had forced FOUR wavefronts per workgroup at every batch size and",
           "undervalued the fused kernel at the large batches (16 utterances, M = 201, N = 4096: 0.521 against the 0.69 below);",
           "its lines are kept as `r04_tune_routing_k{20,80}_f32_w4.jsonl`.", ""]
    for k in (80, 20):
        table(os.path.join(ROOT, "profiles", f"r04_tune_routing_k{k}.jsonl"), out)
    open(os.path.join(ROOT, "profiles", "r04_routing_table.md"), "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
