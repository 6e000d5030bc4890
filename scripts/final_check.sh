#!/bin/bash
# Runs on the GPU box: the whole -m gpu suite, smoke(), one default bench run; summaries land in gpurun_out/.
cd "$GRAFT_REPO_ROOT"
TAG=${1:-final}
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu --durations=6 > gpurun_out/${TAG}_gpu_tests.txt 2>&1
tail -12 gpurun_out/${TAG}_gpu_tests.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 bench.py --steps 3 --warmup 1 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python3 - <<PY
import json
b = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
print("bench:", b["value"], b["result_ok"], "frac", b["roofline"]["frac"], "callers", b["callers"]["find_wall_ms_1_caller"],
      b["callers"]["find_wall_ms_8_callers"], b["callers"]["ratio_8_over_1"], "sharded", [c["wall_ms"] for c in b["sharded"]["cases"]],
      "cpu", b.get("cpu_baseline", {}).get("value"))
PY
