#!/bin/bash
# round 5, probe 10: which counters does this pool's rocprofv3 offer (TLB / L2 / fabric)?
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 -L > gpurun_out/r05_counters_avail.txt 2>&1 || rocprofv3 --list-avail > gpurun_out/r05_counters_avail.txt 2>&1
grep -o -E "\b(TCC|TCP|TCA|UTCL2|UTCL1|TA|TD|GRBM|SQC?)_[A-Z0-9_]+" gpurun_out/r05_counters_avail.txt | sort -u > gpurun_out/r05_counter_names.txt
wc -l gpurun_out/r05_counter_names.txt
grep -i -E "utcl|tlb|mall|dram|EA0_RD|TCC_REQ|TCC_HIT|TCC_MISS|TCP_TCC|TCP_PENDING|TA_BUSY|TCP_TOTAL" gpurun_out/r05_counter_names.txt | head -80
