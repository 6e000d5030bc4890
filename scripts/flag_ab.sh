#!/bin/bash
# developer A/B of compiler scheduling flags: the same sources built with different -mllvm options (build/variants/*.so),
# each timed on the raw PBS micro-benchmark in ONE run; first column of every pair is the shipped library
cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  echo "== shipped"; python3 scripts/bench_pbs.py 5 512,16384 2>&1 | grep count
  echo "== $v"; FHES_LIB=$PWD/build/variants/lib_$v.so python3 scripts/bench_pbs.py 5 512,16384 2>&1 | grep count
done
