#!/bin/bash
# round 5, final records (b): bench lines (three parameter sets), the two-rank rehearsal started by bench.py itself, config 5
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_line.err || { tail -20 gpurun_out/r05_bench_line.err; exit 1; }
timeout -k 10 300 python bench.py --params multibit2 --no-cpu-baseline > gpurun_out/r05_bench_line_multibit2.json 2>> gpurun_out/r05_bench_line.err
timeout -k 10 300 python bench.py --params multibit3 --no-cpu-baseline > gpurun_out/r05_bench_line_multibit3.json 2>> gpurun_out/r05_bench_line.err
FHES_BENCH_REHEARSAL=1 timeout -k 10 400 python bench.py --gpus 2 --steps 2 --no-cpu-baseline > gpurun_out/r05_bench_rehearsal_2ranks.json 2> gpurun_out/r05_bench_rehearsal.err || { tail -20 gpurun_out/r05_bench_rehearsal.err; exit 1; }
python - <<'PY'
import json
for f in ("r05_bench_line", "r05_bench_line_multibit2", "r05_bench_line_multibit3", "r05_bench_rehearsal_2ranks"):
    l = json.loads([x for x in open(f"gpurun_out/{f}.json") if x.startswith("{")][-1])
    r = l["roofline"]
    print(f, {k: l[k] for k in ("value", "n_gpus", "ms_per_step", "finds_per_s", "find_wall_ms_single", "find_wall_ms_single_logdepth", "result_ok")},
          {k: r.get(k) for k in ("bound", "achieved", "peak", "frac", "traffic", "launch_ms", "pbs_per_launch")}, (l.get("sharded") or {}).get("world"))
PY
timeout -k 10 300 python scripts/config5.py 1 all faithful > gpurun_out/r05_config5_g1_faithful.jsonl 2>&1
for g in 1 2 3; do timeout -k 10 120 python scripts/config5.py $g all logdepth > gpurun_out/r05_config5_g${g}_logdepth.jsonl 2>&1; done
cat gpurun_out/r05_config5_g1_faithful.jsonl gpurun_out/r05_config5_g2_logdepth.jsonl | cut -c1-200
