#!/bin/bash
# round 5, probe 13: g = 3 (and g = 2, classic) rate by launch size -- do short launches (one round of 512 workgroups) hold the good mode?
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r05_probe13_launch_size.txt
: > $O
for i in 1 2; do
  timeout -k 10 300 python scripts/bench_pbs.py 0 512,1024,2048,4096,16384,512,1024,2048,4096,16384 multibit3 >> $O 2>&1
done
timeout -k 10 300 python scripts/bench_pbs.py 0 512,1024,2048,16384 multibit2 >> $O 2>&1
timeout -k 10 300 python scripts/bench_pbs.py 0 512,1024,2048,16384 default >> $O 2>&1
cat $O
