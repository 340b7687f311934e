#!/bin/bash
# kernel-trace summary of any python tool (GPU box): bash tools/kt_cmd.sh <outdir> <script.py> [args...]
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o p --output-format csv -- python3 "$@" > "$OUT/run.json" 2> "$OUT/kt.err" || tail -5 "$OUT/kt.err"
python3 tools/trim_stats.py "$OUT/kt/p_kernel_stats.csv" > "$OUT/kernel_stats.csv" 2>/dev/null || true
rm -f "$OUT/kt/p_kernel_trace.csv"
cat "$OUT/run.json"
head -9 "$OUT/kernel_stats.csv" | cut -c1-170
