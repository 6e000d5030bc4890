#!/bin/bash
# developer A/B on the GPU box: the shipped build against another build of the library (FHES_LIB), same run, alternating;
# blind rotate at 64 (one workgroup per CU: the latency regime), 512 and 16 384 ciphertexts
# usage: lib_ab.sh [other.so] [parameter set: default | multibit2 | multibit3] [variant]
cd "$GRAFT_REPO_ROOT"
BASE=${1:-fhe_string_amd/libfhe_base.so}
WHICH=${2:-default}
VAR=${3:-0}
for rep in 1 2; do
  for lib in "$BASE" fhe_string_amd/libfhe_string_amd.so; do
    echo "== $lib"
    FHES_LIB=$PWD/$lib python3 scripts/bench_pbs.py $VAR 64,512,16384 $WHICH 2>&1 | grep -v "^\[build\]"
  done
done
