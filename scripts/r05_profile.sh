#!/bin/bash
# round 5: rocprofv3 passes of the bench command on the final build, one parameter set per call; the raw counter CSVs are
# condensed ON the box (they exceed what gpurun carries back) and only the summaries travel: gpurun_out/profiles_out/
# usage: scripts/r05_profile.sh classic|multibit2|multibit3
set -e
cd "$(dirname "$0")/.."
P=${1:-classic}
if [ "$P" = classic ]; then TAG=r05; EXTRA=""; else TAG=r05_$P; EXTRA="--params $P"; fi
scripts/profile_round.sh $TAG 2 "$EXTRA"
python3 scripts/summarize_profile.py $TAG "$EXTRA" > gpurun_out/prof_$TAG/summary_stdout.txt 2>&1 || { tail -20 gpurun_out/prof_$TAG/summary_stdout.txt; exit 1; }
mkdir -p gpurun_out/profiles_out
cp profiles/${TAG}_* gpurun_out/profiles_out/ 2>/dev/null || true
cp profiles/compute_roofline*.json profiles/pmc_traffic*.json gpurun_out/profiles_out/
cp gpurun_out/prof_$TAG/summary_stdout.txt gpurun_out/profiles_out/${TAG}_summary_stdout.txt
cp gpurun_out/prof_$TAG/bench_line.json gpurun_out/profiles_out/${TAG}_bench_line_under_profiler.json 2>/dev/null || true
# keep the per-kernel statistics verbatim, drop the raw traces and counter dumps
find gpurun_out/prof_$TAG -name '*_kernel_stats.csv' -exec cp {} gpurun_out/profiles_out/${TAG}_kernel_stats.csv \;
rm -rf gpurun_out/prof_$TAG
tail -8 gpurun_out/profiles_out/${TAG}_summary_stdout.txt
