#!/bin/bash
# Round-4 evidence for the task-queue kernels after the first-ticket fix, the static schedule and the 3-tile instance
# (GPU box, from the repo root): bench lines by batch size, kernel trace + PMC of the STFT flow (16 and 2 utterances),
# of the float64 STFT flow and of C3 x 16.   bash tools/r04_profiles_wide.sh
OUT=gpurun_out/r04w
mkdir -p $OUT
: > $OUT/bench_wide.jsonl
for U in 16; do
  python3 bench.py --config STFT --utterances $U --steps 3 --warmup 1 --no-cpu --no-pcie >> $OUT/bench_wide.jsonl 2>> $OUT/bench.err
done
python3 bench.py --config STFT64 --steps 3 --warmup 1 --no-cpu --no-pcie >> $OUT/bench_wide.jsonl 2>> $OUT/bench.err
for U in 16; do
  python3 bench.py --config C3 --utterances $U --steps 2 --warmup 1 --no-cpu --no-pcie >> $OUT/bench_wide.jsonl 2>> $OUT/bench.err
done
echo "bench lines done" >&2
bash tools/prof_bench.sh $OUT/stft16 k_fused_wide 201 4096 150 11008 f32 --config STFT > $OUT/prof_stft16.log 2>&1
bash tools/prof_bench.sh $OUT/stft2 k_fused_wide 201 4096 150 1376 f32 --config STFT --utterances 2 > $OUT/prof_stft2.log 2>&1
bash tools/prof_bench.sh $OUT/stft64f k_fused_wide64 201 4096 150 11008 f64 --config STFT64 > $OUT/prof_stft64f.log 2>&1
bash tools/prof_bench.sh $OUT/c3x16 k_fused_wide64 513 8192 200 11008 f64 --config C3 --utterances 16 > $OUT/prof_c3x16.log 2>&1
echo "profiles done" >&2
