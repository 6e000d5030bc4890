#!/bin/bash
# round 5, probe 4: same-box A/B of the headline -- latency build on / off, round model 1.55 / 1.35
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r05_probe4_ab.txt
: > $out
for cfg in "1.55 256" "1.35 256" "1.35 0" "1.55 256" "1.35 256"; do
  set -- $cfg
  FHES_PACK_FULL_ROUND=$1 FHES_LAT_MAX=$2 timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/tmp_bench.json 2> gpurun_out/tmp_bench.err
  python - "$1" "$2" >> $out <<PY
import json,sys
l=json.loads([x for x in open("gpurun_out/tmp_bench.json") if x.startswith("{")][-1])
print("full_round", sys.argv[1], "lat_max", sys.argv[2], {k:l[k] for k in ("value","ms_per_step","finds_per_s","find_wall_ms_single","find_wall_ms_single_logdepth","result_ok")}, "levels", l["config"]["levels"], l["config"]["pbs_per_level_of_one_find"][:12], "br alone", l["roofline"]["achieved"])
PY
done
cat $out
