#!/bin/bash
# final build of round 5: profile of the bench command, then the bench line that reads its counters
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
scripts/r05_profile.sh classic
echo "profile done" >> gpurun_out/heartbeat.txt
timeout -k 10 300 python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_line.err || { tail -20 gpurun_out/r05_bench_line.err; exit 1; }
python - <<'PY'
import json
l = json.loads([x for x in open("gpurun_out/r05_bench_line.json") if x.startswith("{")][-1])
r = l["roofline"]
print({k: l[k] for k in ("value", "ms_per_step", "finds_per_s", "find_wall_ms_single", "find_wall_ms_single_logdepth", "result_ok")},
      {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "frac", "traffic", "launch_ms", "pbs_per_launch")})
s = json.load(open("gpurun_out/profiles_out/r05_summary.json"))
print(s["duration_cross_check"])
PY
