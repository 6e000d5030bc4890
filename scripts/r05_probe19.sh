#!/bin/bash
# round 5, probe 19: the split multi-bit blind rotate (bundles on the whole device first, then the sequential kernel) -- bits, then time
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python scripts/mb_lat_check.py multibit2 multibit3 > gpurun_out/r05_probe19_split_bits.txt 2>&1 || { tail -20 gpurun_out/r05_probe19_split_bits.txt; exit 1; }
cat gpurun_out/r05_probe19_split_bits.txt
O=gpurun_out/r05_probe19_split_rates.txt
: > $O
for P in multibit2 multibit3; do timeout -k 10 200 python scripts/bench_pbs.py 4,16,4,16 1,4,16,64 $P >> $O 2>&1; done
cat $O
