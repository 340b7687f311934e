#!/bin/bash
# kernel-trace summary of one bench.py invocation (GPU box): bash tools/kt.sh <outdir> [bench.py args...]
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-pcie "$@" > "$OUT/bench.json" 2> "$OUT/kt.err" || tail -5 "$OUT/kt.err"
python3 tools/trim_stats.py "$OUT/kt/p_kernel_stats.csv" > "$OUT/kernel_stats.csv" 2>/dev/null || true
rm -f "$OUT/kt/p_kernel_trace.csv"
head -5 "$OUT/kernel_stats.csv" | cut -c1-160
