#!/bin/bash
# round 4 GPU probe 4: does the g = 3 rate depend on where the key lies?  (pads in front of the engine-layout key; 2 runs each)
mkdir -p gpurun_out
O=gpurun_out/r04_probe4.txt
: > $O
for rep in 1 2; do
  for pad in 0 4096 65536 262144 1048576 2097152 16777216 33554432; do
    echo "-- pad $pad rep $rep" >> $O
    FHES_DEBUG_ADDR=1 FHES_BSK_E_PAD=$pad timeout -k 10 120 python scripts/bench_pbs.py 0 16384 multibit3 >> $O 2>&1
  done
done
grep -E "pad|PBS/s|bsk_e" $O
