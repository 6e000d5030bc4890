#!/bin/bash
# round 4 GPU probe 8: strings per step (the serial levels of a find are 3 bootstraps wide since if_then_else became one lookup per block:
# 3 x 256 = 768 is a round and a half)
mkdir -p gpurun_out
O=gpurun_out/r04_ab_batch_size2.txt
: > $O
for rep in 1 2; do
  for B in 256 512 1024; do
    timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch $B > gpurun_out/_b.json 2> gpurun_out/_b.err || { tail -n 5 gpurun_out/_b.err; exit 1; }
    python - "$B" >> $O <<'PY'
import json, sys
d = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
print("batch", sys.argv[1], "value", d["value"], "ms_per_step", d["ms_per_step"], "finds_per_s", d["finds_per_s"], "br_alone", d["roofline"].get("achieved"), "launches", d["roofline"].get("launches"))
PY
  done
done
cat $O
