#!/bin/bash
# round 5, probe 2: half barriers (LDS flags) in the latency build; ablations
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_pbs.py -x -q -m gpu -k "variants_agree" > gpurun_out/r05_probe2_test.txt 2>&1 || { tail -30 gpurun_out/r05_probe2_test.txt; exit 1; }
tail -3 gpurun_out/r05_probe2_test.txt
FHES_DEV_ABLATIONS=1 timeout -k 10 300 python scripts/bench_pbs.py 3,20,21,22,23,24,25 1,4 > gpurun_out/r05_probe2_lat.txt 2>&1
cat gpurun_out/r05_probe2_lat.txt
