#!/bin/bash
# round 5: GPU suite + headline bench (+ the classic config-5 faithful suite when asked)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
tag=${1:-a}
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gpu_tests_$tag.txt 2>&1 || { tail -40 gpurun_out/r05_gpu_tests_$tag.txt; exit 1; }
tail -4 gpurun_out/r05_gpu_tests_$tag.txt
timeout -k 10 400 python bench.py --steps 5 --warmup 1 > gpurun_out/r05_bench_$tag.json 2> gpurun_out/r05_bench_$tag.err || { tail -20 gpurun_out/r05_bench_$tag.err; exit 1; }
python - <<PY
import json
l=json.loads([x for x in open("gpurun_out/r05_bench_$tag.json") if x.startswith("{")][-1])
print({k:l[k] for k in ("value","ms_per_step","finds_per_s","find_wall_ms_single","find_wall_ms_single_logdepth","result_ok")})
print(l["callers"])
print([ (c["op"][:30], c["wall_ms"], c["result_ok"]) for c in l["sharded"]["cases"]])
PY
