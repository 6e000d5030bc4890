#!/bin/bash
# developer A/B on the GPU box: the shipped build against another build of the library (FHES_LIB), same run, alternating;
# blind rotate at 256 (one workgroup per CU: the latency regime), 512 and 16 384 ciphertexts
cd "$GRAFT_REPO_ROOT"
BASE=${1:-fhe_string_amd/libfhe_base.so}
for rep in 1 2; do
  for lib in "$BASE" fhe_string_amd/libfhe_string_amd.so; do
    echo "== $lib"
    FHES_LIB=$PWD/$lib python3 scripts/bench_pbs.py 5 64,256,512,16384 2>&1 | grep -v "^\[build\]"
  done
done
