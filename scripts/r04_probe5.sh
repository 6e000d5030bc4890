#!/bin/bash
# round 4 GPU probe 5: variant 8 (three workgroups per CU) and its timing-only ablations (13: own words not stored / reloaded,
# 14: also no key loads) against the default kernel
mkdir -p gpurun_out
O=gpurun_out/r04_probe5b.txt
: > $O
for rep in 1; do
  for v in 8 13 14; do
    echo "-- variant $v rep $rep" >> $O
    FHES_DEV_ABLATIONS=1 timeout -k 10 200 python scripts/bench_pbs.py $v 256,768,3072,16384 default >> $O 2>&1 || { tail -30 $O; exit 1; }
  done
done
tail -60 $O
