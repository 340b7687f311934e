#!/bin/bash
# PMC passes over one bench step (separate passes: the TCC block has 4 slots, SQ 8).
# usage (on the GPU box, from the repo root): bash tools/pmc_fused.sh <outdir> [bench args...]
set -e
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" -d "$OUT/$name" -o p --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu "${EXTRA[@]}" > "$OUT/$name.json" 2> "$OUT/$name.err" || { tail -5 "$OUT/$name.err"; return 1; }
}
EXTRA=("$@")
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
