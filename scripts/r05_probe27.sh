#!/bin/bash
# phase stamps of the wave-per-polynomial build: one bootstrap per CU (256) and four per CU (1024)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
FHES_LAB=1 timeout -k 10 300 python scripts/wpp_stamps.py 256 1024 > gpurun_out/r05_probe27_wpp_stamps.txt 2>&1 || { tail -20 gpurun_out/r05_probe27_wpp_stamps.txt; exit 1; }
rm -f gpurun_out/wpp_stamps.bin
cat gpurun_out/r05_probe27_wpp_stamps.txt
