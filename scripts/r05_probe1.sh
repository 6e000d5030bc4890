#!/bin/bash
# round 5, probe 1: the latency build of the classic blind rotate -- bits first, then time per launch size
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_pbs.py -x -q -m gpu -k "variants_agree" > gpurun_out/r05_probe1_test.txt 2>&1 || { tail -30 gpurun_out/r05_probe1_test.txt; exit 1; }
tail -3 gpurun_out/r05_probe1_test.txt
timeout -k 10 300 python scripts/bench_pbs.py 3,2 1,4,64,256 > gpurun_out/r05_probe1_lat.txt 2>&1
cat gpurun_out/r05_probe1_lat.txt
