#!/bin/bash
# round 5, probe 6: server-side serde, device-side clone, key broadcast with the standard-domain words
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_serde.py tests/test_gpu_multirank.py tests/test_gpu_rccl_binding.py tests/test_gpu_api_surface.py -x -q -m gpu > gpurun_out/r05_probe6.txt 2>&1 || { tail -40 gpurun_out/r05_probe6.txt; exit 1; }
tail -4 gpurun_out/r05_probe6.txt
