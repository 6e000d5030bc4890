#!/bin/bash
# PREPARED AT THE END OF ROUND 5 WITHOUT A DEVICE IN REACH (host-side knob of the LAB build: compiled, never run on a device).
# Multi-bit g = 3 lands on 105 / 125 / 143 k PBS/s from process to process at 16 384 bootstraps per launch (TCC_MISS x 3.5 in the slow
# ones), while launches of 4096 measured 139-141 k four times out of four in one process (r05_probe13).  Question: is the
# slow state something a LONG launch drifts into (then a launch boundary every 4096 / 2048 bootstraps keeps a process at ~140 k and the
# product's chunk of 16 384 should shrink for g = 3), or a property of the process (then every chunk size is slow in a slow process)?
# Six processes, each alternating the chunk sizes on the same key.  Gate: chunk 4096 within 3 % across the six processes AND not below
# the best 16 384 figure by more than 3 % -> make it the g = 3 launch size of the product; else record and drop.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r06_probe3_launch_chunk_g3.txt
: > $O
[ -f fhe_string_amd/libfhe_string_amd_lab.so ] || FHES_LAB=1 python -m fhe_string_amd.build > gpurun_out/r06_probe3_build.txt 2>&1 || { tail -20 gpurun_out/r06_probe3_build.txt; exit 1; }
for i in 1 2 3 4 5 6; do
  echo "== process $i" >> $O
  FHES_LAB=1 timeout -k 10 150 python scripts/launch_chunk_probe.py multibit3 2 >> $O 2>&1 || { tail -5 $O; exit 1; }
done
cat $O
