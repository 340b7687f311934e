#!/bin/bash
# Kernel trace + PMC passes over one case of tools/bench_configs.py (GPU box only).
# usage (from the repo root): bash tools/prof_case.sh <outdir> <case> [nopmc]
# The program itself follows `--` (no env/bash hop: rocprofv3's preload initialises the GPU).
OUT=$1; CASE=$2; NOPMC=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
echo "== $CASE kernel trace" >&2
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o p --output-format csv -- python3 tools/bench_configs.py "$CASE" > "$OUT/kt.jsonl" 2> "$OUT/kt.err" || tail -5 "$OUT/kt.err"
python3 tools/trim_stats.py "$OUT/kt/p_kernel_stats.csv" > "$OUT/kernel_stats.csv" 2>/dev/null || true
[ -n "$NOPMC" ] && exit 0
run() { # name counters...
  local name=$1; shift
  echo "== $CASE pmc $name" >&2
  rocprofv3 --pmc "$@" -d "$OUT/$name" -o p --output-format csv -- python3 tools/bench_configs.py "$CASE" > "$OUT/$name.jsonl" 2> "$OUT/$name.err" || tail -3 "$OUT/$name.err"
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.txt"
