#!/bin/bash
# round 5, probe 11: SCAL selection (bits + rates at 1 / 4 / 256 / 16384), then what distinguishes fast from slow g = 3 PROCESSES:
# six processes under rocprofv3, counters of the wide blind-rotate launches (L2 hit / miss / tag stalls, L1 TLB misses, read latency)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_pbs.py -x -q -m gpu > gpurun_out/r05_probe11_test.txt 2>&1 || { tail -30 gpurun_out/r05_probe11_test.txt; exit 1; }
tail -2 gpurun_out/r05_probe11_test.txt
O=gpurun_out/r05_probe11_multibit_rates.txt
: > $O
for P in multibit2 multibit3 multibit2 multibit3 multibit3; do
  timeout -k 10 200 python scripts/bench_pbs.py 0 1,4,256,16384 $P >> $O 2>&1
done
cat $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r05_probe11_pmc.txt
: > $D
for i in 1 2 3 4 5 6; do
  rm -rf /tmp/pmc_$i
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d /tmp/pmc_$i -- python3 scripts/bench_pbs.py 0 16384 multibit3 > /tmp/pmc_$i.log 2>&1 || { tail -5 /tmp/pmc_$i.log >> $D; continue; }
  python3 - $i >> $D <<'PY'
import csv, glob, sys, collections
i = sys.argv[1]
f = glob.glob(f"/tmp/pmc_{i}/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(float); n = 0; dur = 0.0; seen = set()
for r in csv.DictReader(open(f)):
    if "blind_rotate_mb" not in r["Kernel_Name"] or float(r["Grid_Size"]) / float(r["Workgroup_Size"]) < 512: continue
    agg[r["Counter_Name"]] += float(r["Counter_Value"])
    d = r.get("Dispatch_Id", r.get("Dispatch_ID"))
    if d not in seen:
        seen.add(d); n += 1; dur += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
rate = 16384 * n / (dur * 1e-9) if dur else 0
print(f"process {i}: launches {n} avg {dur / max(n, 1) * 1e-6:.2f} ms = {rate:.0f} PBS/s under the profiler; " + ", ".join(f"{k}={v / max(n, 1):.4g}" for k, v in sorted(agg.items())))
PY
  grep "PBS/s" /tmp/pmc_$i.log >> $D
done
cat $D
