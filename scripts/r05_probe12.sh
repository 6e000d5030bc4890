#!/bin/bash
# round 5, probe 12: the same g = 3 key at several addresses inside ONE process (three processes)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r05_probe12_key_placement.txt
: > $O
for i in 1 2 3; do
  echo "== process $i" >> $O
  FHES_LAB=1 FHES_BSK_E_CANDIDATES=6 FHES_BSK_E_HOLES=$((i % 2)) timeout -k 10 300 python scripts/key_placement_probe.py multibit3 >> $O 2>&1
done
cat $O
