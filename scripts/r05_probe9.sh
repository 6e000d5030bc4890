#!/bin/bash
# round 5, probe 9: factored monomial spectra in the multi-bit kernels -- bits, then rates
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_pbs.py -x -q -m gpu > gpurun_out/r05_probe9_test.txt 2>&1 || { tail -30 gpurun_out/r05_probe9_test.txt; exit 1; }
tail -2 gpurun_out/r05_probe9_test.txt
O=gpurun_out/r05_probe9_multibit_rates.txt
: > $O
for P in multibit2 multibit3 multibit2 multibit3; do
  timeout -k 10 200 python scripts/bench_pbs.py 0 1,4,256,16384 $P >> $O 2>&1
done
cat $O
