#!/bin/bash
# round 5, probe 14: latency build, stores interleaved with the last butterflies (lab variant 25) against the default (2)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
FHES_LAB=1 timeout -k 10 300 python -m pytest tests/test_gpu_pbs.py -x -q -m gpu -k "variants_agree" > gpurun_out/r05_probe14_test.txt 2>&1 || { tail -30 gpurun_out/r05_probe14_test.txt; exit 1; }
tail -2 gpurun_out/r05_probe14_test.txt
FHES_LAB=1 timeout -k 10 300 python scripts/bench_pbs.py 2,24,23,2,24,23 1,4,256 > gpurun_out/r05_probe14_lat_ils.txt 2>&1
cat gpurun_out/r05_probe14_lat_ils.txt
