#!/bin/bash
# round 5, probe 20: where the split form stops paying (launch sizes 32..256) against the fused 512-thread latency build
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r05_probe20_split_crossover.txt
: > $O
for P in multibit2 multibit3; do timeout -k 10 300 python scripts/bench_pbs.py 4,16 32,64,96,128,192,256 $P >> $O 2>&1; done
cat $O
