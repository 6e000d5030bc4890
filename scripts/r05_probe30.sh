#!/bin/bash
# automatic choice of the wave-per-polynomial build: raw rates under variant 0, then the bench line with the rule on / off (same box)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r05_probe30_wpp_auto.txt
: > $O
timeout -k 10 300 python scripts/bench_pbs.py 0,3 768,1000,1024,1280,2000,2048,4096 default >> $O 2>&1
for A in 1 0 1 0; do
  echo "FHES_WPP_AUTO=$A" >> $O
  FHES_WPP_AUTO=$A timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"find_wall_ms_single": [0-9.]*' | tr '\n' ' ' >> $O
  echo >> $O
done
cat $O
