#!/bin/bash
# PREPARED AT THE END OF ROUND 5, NOT RUN.  Strings per step of the bench (--batch) against the round grid of the wave-per-polynomial
# build: at 256 strings the 17 three-wide levels of a find are launches of 768 (throughput build, 105 k PBS/s) and the 48 four-wide
# ones launches of 1024 (one round, 122 k); at 512 they become 1536 / 2048, at 1024 3072 / 4096 -- whole rounds of 1024 for both.
# Model from profiles/r05_probe33_wpp2_sizes.txt: +1.5 % at 512, +2 % at 1024 on `value`.  Same box, same process order.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r06_probe2_batch_ab.txt
: > $O
for B in 256 512 1024 256; do
  echo "== --batch $B" >> $O
  timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch $B 2>> $O | python -c "
import json, sys
for line in sys.stdin:
    if line.lstrip().startswith('{'):
        j = json.loads(line)
        print(json.dumps({k: j.get(k) for k in ('value', 'ms_per_step', 'finds_per_s', 'result_ok')}), 'launch_ms', j['roofline'].get('launch_ms'), 'pbs_per_launch', j['roofline'].get('pbs_per_launch'))
" >> $O
done
cat $O
