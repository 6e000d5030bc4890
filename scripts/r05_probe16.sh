#!/bin/bash
# round 5, probe 16: lone rounds of the throughput kernel (257..512 bootstraps) under the three wave-priority policies
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
FHES_LAB=1 timeout -k 10 400 python scripts/bench_pbs.py 3,5,6,3,5,6 320,512,768,1024 > gpurun_out/r05_probe16_lone_rounds.txt 2>&1
cat gpurun_out/r05_probe16_lone_rounds.txt
