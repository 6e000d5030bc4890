#!/bin/bash
# wave-per-polynomial build: what the table requests and the GGSW stream cost (timing-only ablations 28 / 29, lab build)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r05_probe28_wpp_ablations.txt
FHES_LAB=1 FHES_DEV_ABLATIONS=1 timeout -k 10 400 python scripts/bench_pbs.py 3,17,28,29 256,1024,4096 default > $O 2>&1 || { tail -20 $O; exit 1; }
cat $O
