#!/bin/bash
# round 4 GPU probe 2: multi-bit g = 2 layout A/B (alternating), multirank key broadcast with the interleaved g = 3 key
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r04_probe2.txt
: > $O
timeout -k 10 400 python -m pytest tests/test_gpu_pbs.py tests/test_gpu_multirank.py -x -q -m gpu >> $O 2>&1 || { tail -30 $O; exit 1; }
FHES_MB_INTERLEAVE=1 timeout -k 10 300 python -m pytest tests/test_gpu_pbs.py -x -q -m gpu -k "multibit" >> $O 2>&1 || { tail -30 $O; exit 1; }
FHES_MB_INTERLEAVE=0 timeout -k 10 300 python -m pytest tests/test_gpu_pbs.py -x -q -m gpu -k "multibit" >> $O 2>&1 || { tail -30 $O; exit 1; }
echo "== multibit2 throughput, layouts alternating (0 = plain, 1 = interleaved)" >> $O
for rep in 1 2 3; do
  for ilv in 0 1; do
    echo "-- FHES_MB_INTERLEAVE=$ilv rep $rep" >> $O
    FHES_MB_INTERLEAVE=$ilv timeout -k 10 120 python scripts/bench_pbs.py 0 256,16384 multibit2 >> $O 2>&1
  done
done
echo "== multibit3, default layout (interleaved) and plain" >> $O
timeout -k 10 120 python scripts/bench_pbs.py 0 256,16384 multibit3 >> $O 2>&1
FHES_MB_INTERLEAVE=0 timeout -k 10 120 python scripts/bench_pbs.py 0 256,16384 multibit3 >> $O 2>&1
tail -40 $O
