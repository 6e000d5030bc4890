#!/bin/bash
# wave-per-polynomial build, final form: the variant tests of both libraries, then rates by launch size against the throughput build
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_pbs.py tests/test_gpu_param_shapes.py -q -m gpu -x -k "variants or wave_per or generic_kernel_equals" > gpurun_out/r05_probe29_tests.txt 2>&1 || { tail -30 gpurun_out/r05_probe29_tests.txt; exit 1; }
tail -3 gpurun_out/r05_probe29_tests.txt
FHES_LAB=1 timeout -k 10 600 python -m pytest tests/test_gpu_pbs.py -q -m gpu -x -k "variants" > gpurun_out/r05_probe29_tests_lab.txt 2>&1 || { tail -30 gpurun_out/r05_probe29_tests_lab.txt; exit 1; }
tail -3 gpurun_out/r05_probe29_tests_lab.txt
O=gpurun_out/r05_probe29_wpp_rates.txt
: > $O
timeout -k 10 500 python scripts/bench_pbs.py 3,17,3,17 256,512,768,1024,1536,2048,4096,16384 default >> $O 2>&1
cat $O
