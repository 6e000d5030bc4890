#!/bin/bash
# round 5, probe 8: what bounds the multi-bit kernels?  timing-only probes (lab build): monomial factors from a table cut down
# to 64 entries (the gathers hit close by), the key of ONE group for every step (the key stream hits close by), both
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r05_probe8_multibit_bounds.txt
: > $O
for P in multibit2 multibit3; do
  for cfg in "-1 -1" "63 -1" "-1 0" "63 0" "-1 -1"; do
    set -- $cfg
    echo "== $P root_mask=$1 group_mask=$2" >> $O
    FHES_LAB=1 FHES_MB_ROOT_MASK=$1 FHES_MB_GROUP_MASK=$2 timeout -k 10 200 python scripts/bench_pbs.py 0 16384 $P >> $O 2>&1
  done
done
cat $O
