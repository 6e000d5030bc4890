#!/bin/bash
# wave-per-polynomial build (variant 17): bits against the oracle, then its rate against the throughput build by launch size
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python scripts/mb_lat_check.py default > gpurun_out/r05_probe26_bits.txt 2>&1 || { tail -20 gpurun_out/r05_probe26_bits.txt; exit 1; }
cat gpurun_out/r05_probe26_bits.txt
O=gpurun_out/r05_probe26_wpp_rates.txt
: > $O
timeout -k 10 400 python scripts/bench_pbs.py 3,17,3,17 256,1024,4096,16384 default >> $O 2>&1
cat $O
