#!/bin/bash
# round 5: where the GPU suite spends its time
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=30 > gpurun_out/r05_gpu_tests_durations.txt 2>&1 || { tail -40 gpurun_out/r05_gpu_tests_durations.txt; exit 1; }
grep -A34 "slowest" gpurun_out/r05_gpu_tests_durations.txt | head -40; tail -2 gpurun_out/r05_gpu_tests_durations.txt
