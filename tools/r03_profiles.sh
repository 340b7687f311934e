#!/bin/bash
# Round-3 evidence run (GPU box, from the repo root): bench lines of every preset, kernel trace + PMC of the headline
# (C2) and of the STFT flow, the N = 2 rehearsal of bench.py's own rank start.   bash tools/r03_profiles.sh
OUT=gpurun_out/r03
mkdir -p $OUT
: > $OUT/bench_all_configs.jsonl
for c in C2 C1 C3 C4 C5 C5_513 STFT; do
  python3 bench.py --config $c --steps 5 --warmup 2 --no-cpu >> $OUT/bench_all_configs.jsonl 2>> $OUT/bench.err
done
python3 bench.py --config STFT --utterances 64 --steps 3 --warmup 1 --no-cpu >> $OUT/bench_all_configs.jsonl 2>> $OUT/bench.err
python3 bench.py --config STFT --utterances 1 --steps 10 --warmup 3 --no-cpu >> $OUT/bench_all_configs.jsonl 2>> $OUT/bench.err
python3 bench.py --config C2 --utterances 1 --steps 10 --warmup 3 --no-cpu >> $OUT/bench_all_configs.jsonl 2>> $OUT/bench.err
python3 bench.py --config C3 --utterances 16 --steps 2 --warmup 1 --no-cpu >> $OUT/bench_all_configs.jsonl 2>> $OUT/bench.err
echo "bench lines done" >&2
python3 bench.py --gpus 2 --same-device --dist-backend gloo --config C4 --steps 3 --warmup 1 --no-cpu --no-pcie > $OUT/bench_c4_2ranks_one_card.json 2>> $OUT/bench.err
echo "2-rank rehearsal done" >&2
bash tools/prof_bench.sh $OUT/c2 k_fused_all 25 4096 100 176128 f64 --config C2 > $OUT/prof_c2.log 2>&1
bash tools/prof_bench.sh $OUT/stft k_fused_wide 201 4096 150 11008 f32 --config STFT > $OUT/prof_stft.log 2>&1
echo "profiles done" >&2
