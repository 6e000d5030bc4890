#!/bin/bash
# developer probe: SQ counters of the blind-rotate kernel on the raw PBS micro-benchmark
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
VAR=${1:-3}; CNT=${2:-512}
OUT=gpurun_out/pmc_probe_${VAR}_${CNT}
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p1 -- python3 scripts/bench_pbs.py $VAR $CNT > $OUT/log1.txt 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/p2 -- python3 scripts/bench_pbs.py $VAR $CNT > $OUT/log2.txt 2>&1
python3 - <<PY
import csv, glob, collections
for sub in ("p1","p2"):
    f=glob.glob("$OUT/%s/*/*_counter_collection.csv"%sub)[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "blind_rotate" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
    for k,v in agg.items():
        val=v[-1]
        print("%-26s %16.0f   (kernel %.3f ms, %d dispatches)"%(k,val[0],val[1]/1e6,len(v)))
PY
