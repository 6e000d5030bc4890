#!/bin/bash
# round 4 GPU probe: bit-exactness of the new kernels, then same-run A/B timings (alternating)
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r04_probe.txt
: > $O
timeout -k 10 500 python -m pytest tests/test_gpu_param_shapes.py tests/test_gpu_pbs.py -x -q -m gpu >> $O 2>&1 || { tail -30 $O; exit 1; }
FHES_MB_INTERLEAVE=1 timeout -k 10 300 python -m pytest tests/test_gpu_pbs.py -x -q -m gpu -k "multibit" >> $O 2>&1 || { tail -30 $O; exit 1; }
echo "== shapes (big kernel on)" >> $O
timeout -k 10 200 python scripts/bench_shapes.py >> $O 2>&1
echo "== multibit3 throughput, layouts alternating (0 = plain, 1 = interleaved)" >> $O
for rep in 1 2 3; do
  for ilv in 0 1; do
    echo "-- FHES_MB_INTERLEAVE=$ilv rep $rep" >> $O
    FHES_MB_INTERLEAVE=$ilv timeout -k 10 120 python scripts/bench_pbs.py 0 4096,16384 multibit3 >> $O 2>&1
  done
done
echo "== multibit2 / default, for reference" >> $O
timeout -k 10 120 python scripts/bench_pbs.py 0 16384 multibit2 >> $O 2>&1
timeout -k 10 120 python scripts/bench_pbs.py 0 16384 default >> $O 2>&1
tail -60 $O
