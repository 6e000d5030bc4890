#!/bin/bash
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python scripts/config3.py > gpurun_out/r05_config3.jsonl 2>&1 || { tail gpurun_out/r05_config3.jsonl; exit 1; }
cat gpurun_out/r05_config3.jsonl
