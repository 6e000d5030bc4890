#!/bin/bash
# round 5, probe 7: config 5 at full size, word for word and block by block
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_config5_fullsize.py -x -q -s -m gpu > gpurun_out/r05_probe7_config5_fullsize.txt 2>&1 || { tail -40 gpurun_out/r05_probe7_config5_fullsize.txt; exit 1; }
grep -E "config5 multibit2|passed|failed" gpurun_out/r05_probe7_config5_fullsize.txt
