#!/bin/bash
# Round-3 probe (runs on the GPU box): variant parity test, then A/B of blind-rotate variants in ONE run.
# usage: r03_probe.sh "<variants>" "<sizes>" [tag]
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
TAG=${3:-a}
timeout -k 10 300 python3 -m pytest tests/test_gpu_pbs.py -x -q -m gpu -k "variants_agree or blind_rotate" 2>&1 | tail -3 | tee gpurun_out/r03_${TAG}_test.txt
timeout -k 10 500 python3 scripts/bench_pbs.py ${1:-5,8,5,8} ${2:-16384} 2>&1 | tee gpurun_out/r03_${TAG}_ab.txt
