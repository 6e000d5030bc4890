#!/bin/bash
# two bootstraps per workgroup (variant 30) against the throughput build (3) by launch size: where does which win?
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r05_probe33_wpp2_sizes.txt
FHES_LAB=1 timeout -k 10 500 python scripts/bench_pbs.py 3,30,3,30 288,320,384,448,512,576,640,768,896,1024,1152,1280,1536,1792,2048,2560,3072,3840,4096 default > $O 2>&1
python - <<'PY'
import re, collections
t = collections.defaultdict(lambda: collections.defaultdict(list))
for l in open("gpurun_out/r05_probe33_wpp2_sizes.txt"):
    m = re.search(r"variant=(\d+) count=\s*(\d+) wall=\s*([\d.]+) ms", l)
    if m: t[int(m.group(2))][int(m.group(1))].append(float(m.group(3)))
for c in sorted(t):
    a, b = min(t[c][3]), min(t[c][30])
    print(f"{c:6d}  v5 {a:7.3f} ms  wpp2 {b:7.3f} ms  ratio {b / a:.3f}")
PY
