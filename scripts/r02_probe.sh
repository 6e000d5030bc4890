#!/bin/bash
# Round-2 probe (runs on the GPU box): kernel-variant parity, A/B of the throughput blind-rotate
# variants, instruction issue costs + the clock held under load, SQ counter passes.
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_gpu_pbs.py -x -q -m gpu -k "variants_agree" 2>&1 | tail -3
timeout -k 10 300 python3 scripts/bench_pbs.py 4,5,4,5 16384 2>&1 | tee gpurun_out/r02_ab.txt
timeout -k 10 120 ./scripts/micro/f64_rate > gpurun_out/r02_f64_rate.jsonl
tail -5 gpurun_out/r02_f64_rate.jsonl
timeout -k 10 400 bash scripts/pmc_probe.sh ${1:-5} ${2:-16384} | tee gpurun_out/r02_pmc_probe_v${1:-5}.txt
