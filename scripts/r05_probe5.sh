#!/bin/bash
# round 5, probe 5: product and lab builds -- variant tests
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_pbs.py tests/test_gpu_param_shapes.py -x -q -m gpu > gpurun_out/r05_probe5_product.txt 2>&1 || { tail -30 gpurun_out/r05_probe5_product.txt; exit 1; }
tail -3 gpurun_out/r05_probe5_product.txt
FHES_LAB=1 timeout -k 10 300 python -m pytest tests/test_gpu_pbs.py -x -q -m gpu -k "variants_agree" > gpurun_out/r05_probe5_lab.txt 2>&1 || { tail -30 gpurun_out/r05_probe5_lab.txt; exit 1; }
tail -3 gpurun_out/r05_probe5_lab.txt
FHES_LAB=1 FHES_DEV_ABLATIONS=1 timeout -k 10 300 python scripts/bench_pbs.py 3,2,20,21,22,23,24,25 1,4 > gpurun_out/r05_probe5_lat.txt 2>&1
cat gpurun_out/r05_probe5_lat.txt
