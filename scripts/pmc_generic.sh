#!/bin/bash
# developer probe: arbitrary PMC counters of the blind-rotate kernels on the raw PBS micro-benchmark
# usage: pmc_generic.sh "<counters>" <variant> <count> [params]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CNTRS="$1"; VAR=${2:-0}; CNT=${3:-4096}; PAR=${4:-default}
OUT=gpurun_out/pmc_generic
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc $CNTRS --kernel-trace --output-format csv -d $OUT/p -- python3 scripts/bench_pbs.py $VAR $CNT $PAR > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
f=glob.glob("$OUT/p/*/*_counter_collection.csv")[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "blind_rotate" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
for k,v in sorted(agg.items()):
    val=v[-1]
    print("%-26s %18.0f   (kernel %.3f ms, %d dispatches)"%(k,val[0],val[1]/1e6,len(v)))
PY
