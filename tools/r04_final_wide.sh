#!/bin/bash
# Round-4 final lines of the task-queue kernels (GPU box, from the repo root): tests, soak, bench lines by batch size, the
# default call.   bash tools/r04_final_wide.sh
OUT=gpurun_out/r04w3
mkdir -p $OUT
python3 -m pytest tests/test_gpu_wide.py tests/test_gpu_wide64.py tests/test_gpu_routed_sizes.py tests/test_gpu_pipeline.py -q 2>&1 | tail -3 || exit 1
timeout -k 10 600 python3 tools/soak_wide.py 300 120 > $OUT/soak.txt 2>&1; tail -2 $OUT/soak.txt
: > $OUT/bench_wide.jsonl
for U in 1 2 3 4 5 6 7 8 9 10 11 12 16 32 64; do
  python3 bench.py --config STFT --utterances $U --steps 4 --warmup 1 --no-cpu --no-pcie >> $OUT/bench_wide.jsonl 2>> $OUT/err.log
done
python3 bench.py --config STFT64 --steps 3 --warmup 1 --no-cpu --no-pcie >> $OUT/bench_wide.jsonl 2>> $OUT/err.log
for U in 1 2 4 6 16; do
  python3 bench.py --config C3 --utterances $U --steps 2 --warmup 1 --no-cpu --no-pcie >> $OUT/bench_wide.jsonl 2>> $OUT/err.log
done
python3 tools/bench_default_call_stft.py 8 16 64 > $OUT/default_call_wide.jsonl
python3 - <<PY
import json
for l in open("$OUT/bench_wide.jsonl"):
    r=json.loads(l); print(r["config"]["workload"][:12], round(r["ms_per_step"],2), round(r["value"]), round(r["roofline"]["frac"],4), r["config"].get("kernel","")[:14])
for l in open("$OUT/default_call_wide.jsonl"):
    r=json.loads(l); print(r["utterances"], r["tol_0"]["ms"], r["tol_0.0001"]["ms"], r["tol_0.0001"]["launches"], r["with_tests_over_without"])
PY
