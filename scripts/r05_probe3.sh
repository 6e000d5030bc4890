#!/bin/bash
# round 5, probe 3: phase stamps of the latency build
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python scripts/lat_stamps.py 4 26 > gpurun_out/r05_probe3_stamps.txt 2>&1 || { tail -30 gpurun_out/r05_probe3_stamps.txt; exit 1; }
cat gpurun_out/r05_probe3_stamps.txt
