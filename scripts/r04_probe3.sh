#!/bin/bash
# round 4 GPU probe 3: the N = 8192 kernel with LDS twiddle tables (bit-exactness first), then the g = 3 layout A/B once more, 5 alternations
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r04_probe3.txt
: > $O
timeout -k 10 500 python -m pytest tests/test_gpu_param_shapes.py tests/test_gpu_api_surface.py -x -q -m gpu >> $O 2>&1 || { tail -30 $O; exit 1; }
echo "== shapes" >> $O
timeout -k 10 200 python scripts/bench_shapes.py >> $O 2>&1
echo "== multibit3 throughput, layouts alternating (0 = plain, 1 = interleaved)" >> $O
for rep in 1 2 3 4 5; do
  for ilv in 0 1; do
    echo "-- FHES_MB_INTERLEAVE=$ilv rep $rep" >> $O
    FHES_MB_INTERLEAVE=$ilv timeout -k 10 120 python scripts/bench_pbs.py 0 16384 multibit3 >> $O 2>&1
  done
done
tail -40 $O
