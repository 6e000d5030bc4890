#!/bin/bash
# GPU fuzz of the final build; every campaign appends its progress lines to the record as it goes (a run behind a pipe looks hung)
mkdir -p gpurun_out
O=gpurun_out/r05_gpu_fuzz.txt
: > $O
timeout -k 10 400 python scripts/gpu_fuzz.py 60 91 classic >> $O 2>&1 || { tail -n 5 $O; exit 1; }
timeout -k 10 260 python scripts/gpu_fuzz.py 30 92 multibit3 >> $O 2>&1 || { tail -n 5 $O; exit 1; }
timeout -k 10 200 python scripts/gpu_fuzz.py 30 93 multibit2 >> $O 2>&1 || { tail -n 5 $O; exit 1; }
grep -c "mismatch" $O; grep "mismatch" $O | grep -v " 0 mismatch" | head -5; tail -n 2 $O
