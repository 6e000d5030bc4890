#!/bin/bash
# developer A/B: bench.py with each blind-rotate variant given on the command line, twice
cd "$GRAFT_REPO_ROOT"
for v in "$@" "$@"; do
  python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --variant $v 2>/dev/null > gpurun_out/ab_$v.json
  python3 -c "
import json
d=json.load(open('gpurun_out/ab_$v.json'))
print('variant $v', d['value'], d['roofline']['frac'], d['find_wall_ms_single'])"
done
