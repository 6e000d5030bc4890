#!/bin/bash
# Round-2 probe (runs on the GPU box): GPU test-suite, A/B of the throughput blind-rotate variants,
# SQ counter passes of one variant.
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5
timeout -k 10 300 python3 scripts/bench_pbs.py ${3:-5,6,7,5,6,7} 16384 2>&1 | tee gpurun_out/r02_ab.txt
timeout -k 10 400 bash scripts/pmc_probe.sh ${1:-5} ${2:-16384} | tee gpurun_out/r02_pmc_probe_v${1:-5}.txt
