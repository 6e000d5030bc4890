#!/bin/bash
# Runs on the GPU box (via gpurun).  Three rocprofv3 passes of the same bench command:
#   1. --kernel-trace --stats            per-kernel durations
#   2. --pmc FETCH_SIZE                  HBM/fabric read bytes (own pass, see MI355X_MICROARCH.md)
#   3. --pmc WRITE_SIZE                  HBM/fabric write bytes (own pass)
# Results land in gpurun_out/prof_<tag>/ ; scripts/summarize_profile.py condenses them into profiles/.
set -e
TAG=${1:-r01}
STEPS=${2:-2}
EXTRA=${3:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
CMD="python3 bench.py --steps $STEPS --warmup 1 --no-cpu-baseline $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/bench_stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- $CMD > "$OUT/bench_write.log" 2>&1
grep -o '{"metric.*' "$OUT/bench_stats.log" > "$OUT/bench_line.json" || true
find "$OUT" -name '*.csv' | head -20
