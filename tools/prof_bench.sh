#!/bin/bash
# rocprofv3 evidence for one bench.py workload (GPU box only; from the repo root):
#   bash tools/prof_bench.sh <outdir> <kernel tag> <M> <N> <K> <frames> <dtype> [bench.py args...]
# -> <outdir>/kernel_stats.csv (--kernel-trace --stats of the same command the bench line comes from),
#    <outdir>/pmc_summary.txt + pmc.json (separate --pmc passes; the json is what bench.py reads for
#    roofline.traffic), <outdir>/bench.json (the bench line of the traced run).
# The program itself follows `--` (no env/bash hop: rocprofv3's preload initialises the GPU).
OUT=$1; TAG=$2; M=$3; N=$4; K=$5; FR=$6; DT=$7; shift 7
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
EXTRA=("$@")
echo "== kernel trace: bench.py ${EXTRA[*]}" >&2
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-pcie "${EXTRA[@]}" > "$OUT/bench.json" 2> "$OUT/kt.err" || tail -5 "$OUT/kt.err"
python3 tools/trim_stats.py "$OUT/kt/p_kernel_stats.csv" > "$OUT/kernel_stats.csv" 2>/dev/null || true
rm -f "$OUT/kt/p_kernel_trace.csv"
run() { # name counters...
  local name=$1; shift
  echo "== pmc $name" >&2
  rocprofv3 --pmc "$@" -d "$OUT/$name" -o p --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-pcie "${EXTRA[@]}" > "$OUT/$name.json" 2> "$OUT/$name.err" || tail -3 "$OUT/$name.err"
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 tools/pmc_summary.py "$OUT" --json "$OUT/pmc.json" "$M" "$N" "$K" "$FR" "$DT" "$TAG" > "$OUT/pmc_summary.txt"
