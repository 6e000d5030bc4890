#!/bin/bash
# round 5, probe 22: same-box A/B of the multi-bit bench lines with the split form on (default) and off
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r05_probe22_split_ab.txt
: > $O
for cfg in "multibit2 0" "multibit2 128" "multibit3 0" "multibit3 192" "multibit2 0" "multibit2 128"; do
  set -- $cfg
  FHES_MB_SPLIT_MAX=$2 timeout -k 10 300 python bench.py --params $1 --steps 4 --no-cpu-baseline > gpurun_out/tmp_bench.json 2> gpurun_out/tmp_bench.err
  python - "$1" "$2" >> $O <<'PY'
import json, sys
l = json.loads([x for x in open("gpurun_out/tmp_bench.json") if x.startswith("{")][-1])
print(sys.argv[1], "split_max", sys.argv[2], {k: l[k] for k in ("value", "finds_per_s", "find_wall_ms_single", "find_wall_ms_single_logdepth", "result_ok")}, "br alone", l["roofline"]["achieved_pbs_per_s"])
PY
done
cat $O
