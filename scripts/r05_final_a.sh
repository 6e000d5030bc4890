#!/bin/bash
# round 5, final records (a): the GPU suite and the smoke
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gpu_tests.txt 2>&1 || { tail -40 gpurun_out/r05_gpu_tests.txt; exit 1; }
tail -3 gpurun_out/r05_gpu_tests.txt
python __graft_entry__.py smoke > gpurun_out/r05_smoke.txt 2>&1 || { tail gpurun_out/r05_smoke.txt; exit 1; }
tail -1 gpurun_out/r05_smoke.txt
