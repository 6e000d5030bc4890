#!/bin/bash
# Round-2 probe (runs on the GPU box): instruction issue costs + the clock held under load,
# then the SQ counter passes of the throughput blind-rotate kernel on a 16384-ciphertext launch.
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
./scripts/micro/f64_rate > gpurun_out/r02_f64_rate.jsonl
cat gpurun_out/r02_f64_rate.jsonl | head -12
bash scripts/pmc_probe.sh ${1:-4} ${2:-16384} | tee gpurun_out/r02_pmc_probe.txt
