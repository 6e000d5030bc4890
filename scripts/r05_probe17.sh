#!/bin/bash
# round 5, probe 17: the 512-thread latency build of the multi-bit blind rotate -- bits, then time per launch size, with and without the L2 touch
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python scripts/mb_lat_check.py multibit2 multibit3 > gpurun_out/r05_probe17_mb_lat_bits.txt 2>&1 || { tail -20 gpurun_out/r05_probe17_mb_lat_bits.txt; exit 1; }
cat gpurun_out/r05_probe17_mb_lat_bits.txt
O=gpurun_out/r05_probe17_mb_lat_rates.txt
: > $O
for P in multibit2 multibit3; do
  echo "== $P, L2 touch on (variant 4) against the 256-thread build (variant 0)" >> $O
  FHES_LAB=1 timeout -k 10 200 python scripts/bench_pbs.py 0,4,4 1,4,256 $P >> $O 2>&1
  echo "== $P, L2 touch off" >> $O
  FHES_LAB=1 FHES_MB_LAT_NOPF=1 timeout -k 10 200 python scripts/bench_pbs.py 4,4 1,4,256 $P >> $O 2>&1
done
cat $O
