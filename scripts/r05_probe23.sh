#!/bin/bash
# round 5, probe 23: is the box slow, or the build?  classic and multi-bit g = 2 in one call
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r05_probe23_box_check.txt
: > $O
for P in classic multibit2; do
  timeout -k 10 300 python bench.py --params $P --steps 3 --no-cpu-baseline > gpurun_out/tmp_bench.json 2> gpurun_out/tmp_bench.err
  python - "$P" >> $O <<'PY'
import json, sys
l = json.loads([x for x in open("gpurun_out/tmp_bench.json") if x.startswith("{")][-1])
print(sys.argv[1], {k: l[k] for k in ("value", "find_wall_ms_single")}, "br alone", l["roofline"]["achieved_pbs_per_s"], "clock", (l["roofline"].get("valu_issue_at_measured_slot_costs") or {}).get("clock_ghz_under_load"))
PY
done
timeout -k 10 200 python scripts/bench_pbs.py 0 16384 default >> $O 2>&1
timeout -k 10 200 python scripts/bench_pbs.py 0 16384 multibit2 >> $O 2>&1
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -4 >> $O || true
cat $O
