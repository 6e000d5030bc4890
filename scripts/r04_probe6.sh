#!/bin/bash
# round 4 GPU probe 6: the block tests against the restatement and the bench line with the opt-in sharded cases
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ref_circuits.py -x -q -m gpu 2>&1 | tail -n 3
python bench.py --steps 5 --warmup 1 > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_bench_line.json").read().strip().splitlines()[-1])
print(d["value"], d["find_wall_ms_single"], d["find_wall_ms_single_logdepth"])
for c in d["sharded"]["cases"]:
    print(c)
PY
