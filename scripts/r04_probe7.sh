#!/bin/bash
# round 4 GPU probe 7: BASELINE config 5 (512-character suite) with the opt-in forms on the three parameter sets, after the count scans,
# the threshold lookups and the launch packing
mkdir -p gpurun_out
for g in 1 2 3; do
  timeout -k 10 280 python scripts/config5.py $g all logdepth > gpurun_out/r04_config5_g${g}_logdepth.jsonl 2>&1 || { tail -n 20 gpurun_out/r04_config5_g${g}_logdepth.jsonl; exit 1; }
  cat gpurun_out/r04_config5_g${g}_logdepth.jsonl
done
