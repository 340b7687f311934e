#!/bin/bash
# Round-4 evidence run (GPU box, from the repo root): bench lines of every preset, kernel trace + PMC of the headline
# (C2), of C5 and of the STFT flow.   bash tools/r04_profiles.sh
OUT=gpurun_out/r04p
mkdir -p $OUT
: > $OUT/bench_all_configs.jsonl
for c in C2 C1 C3 C4 C5 C5_513 STFT; do
  python3 bench.py --config $c --steps 5 --warmup 2 --no-cpu >> $OUT/bench_all_configs.jsonl 2>> $OUT/bench.err
done
python3 bench.py --config STFT --utterances 64 --steps 3 --warmup 1 --no-cpu >> $OUT/bench_all_configs.jsonl 2>> $OUT/bench.err
python3 bench.py --config STFT --utterances 1 --steps 10 --warmup 3 --no-cpu >> $OUT/bench_all_configs.jsonl 2>> $OUT/bench.err
python3 bench.py --config C2 --utterances 1 --steps 10 --warmup 3 --no-cpu >> $OUT/bench_all_configs.jsonl 2>> $OUT/bench.err
python3 bench.py --config C3 --utterances 16 --steps 2 --warmup 1 --no-cpu >> $OUT/bench_all_configs.jsonl 2>> $OUT/bench.err
python3 bench.py --config STFT64 --steps 3 --warmup 1 --no-cpu >> $OUT/bench_all_configs.jsonl 2>> $OUT/bench.err
python3 bench.py --config C2 --pair-tiles --steps 5 --warmup 2 --no-cpu >> $OUT/bench_all_configs.jsonl 2>> $OUT/bench.err
echo "bench lines done" >&2
bash tools/prof_bench.sh $OUT/c2 k_fused_all 25 4096 100 176128 f64 --config C2 > $OUT/prof_c2.log 2>&1
bash tools/prof_bench.sh $OUT/c5 k_fused_all 25 16384 100 11008 f64 --config C5 > $OUT/prof_c5.log 2>&1
echo "profiles done" >&2
