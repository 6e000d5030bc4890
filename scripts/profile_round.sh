#!/bin/bash
# Runs on the GPU box (via gpurun).  rocprofv3 passes of the SAME bench command:
#   1. --kernel-trace --stats            per-kernel durations
#   2. --pmc FETCH_SIZE                  HBM/fabric read bytes (own pass, see MI355X_MICROARCH.md)
#   3. --pmc WRITE_SIZE                  HBM/fabric write bytes (own pass)
#   4. --pmc SQ set A / 5. SQ set B      issue / wait / LDS counters of the compute roofline
#   6. --pmc TCC_*                       L2 hit rate and fabric requests of the key stream
# plus the instruction issue-cost micro-benchmark (scripts/micro/f64_rate).
# Results land in gpurun_out/prof_<tag>/ ; scripts/summarize_profile.py condenses them into profiles/.
set -e
TAG=${1:-r02}
STEPS=${2:-2}
EXTRA=${3:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
CMD="python3 bench.py --steps $STEPS --warmup 1 --no-cpu-baseline $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/bench_stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- $CMD > "$OUT/bench_write.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/sqa" -- $CMD > "$OUT/bench_sqa.log" 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$OUT/sqb" -- $CMD > "$OUT/bench_sqb.log" 2>&1
# 6. L2 (TCC) hit / miss / fabric requests: where the key stream is served from (round 5)
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d "$OUT/tcc" -- $CMD > "$OUT/bench_tcc.log" 2>&1 || echo "tcc pass failed (counters unavailable?)"
grep -o '{"metric.*' "$OUT/bench_stats.log" > "$OUT/bench_line.json" || true
if [ -x scripts/micro/f64_rate ]; then ./scripts/micro/f64_rate > "$OUT/issue_rates.jsonl"; fi
find "$OUT" -name '*.csv' | head -20
