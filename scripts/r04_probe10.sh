#!/bin/bash
# round 4 GPU probe 10: the faithful (reference-order) circuits after the depth work on counters and comparisons: config 5 split / replace /
# find / rfind / trim on the classic set (one GPU), and config 3's faithful split through the sharded entry point
mkdir -p gpurun_out
timeout -k 10 400 python scripts/config5.py 1 all faithful > gpurun_out/r04_config5_g1_faithful.jsonl 2>&1 || { tail -n 20 gpurun_out/r04_config5_g1_faithful.jsonl; exit 1; }
cat gpurun_out/r04_config5_g1_faithful.jsonl
