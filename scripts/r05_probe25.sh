#!/bin/bash
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python scripts/mb_lat_check.py multibit2 multibit3 > gpurun_out/r05_probe25_bits.txt 2>&1 || { tail -20 gpurun_out/r05_probe25_bits.txt; exit 1; }
grep "variant 16" gpurun_out/r05_probe25_bits.txt
O=gpurun_out/r05_probe25_bundle_threads.txt
: > $O
for P in multibit2 multibit3; do timeout -k 10 200 python scripts/bench_pbs.py 16,16 1,4,16,64,128 $P >> $O 2>&1; done
cat $O
