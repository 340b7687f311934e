set -o pipefail
timeout -k 10 300 python tools/wide64_check.py > gpurun_out/w64_check.log 2>&1; rc=$?; echo exit=$rc >> gpurun_out/w64_check.log
if grep -q "Memory access fault" gpurun_out/w64_check.log; then tail -5 gpurun_out/w64_check.log; exit 1; fi
if [ $rc -ne 0 ]; then tail -25 gpurun_out/w64_check.log; exit 1; fi
cd tools/ubench/bin && (timeout -k 10 100 ./wide64_bench 16 0 0 6 && timeout -k 10 100 ./wide64_bench_t 16 0 0 6 && timeout -k 10 100 ./wide64_bench 1 0 0 30) > ../../../gpurun_out/w64_ablate.log 2>&1
cd ../../..; grep -v "^reduce\|^iteration\|^M=" gpurun_out/w64_ablate.log; tail -4 gpurun_out/w64_check.log
