#!/bin/bash
# final records of round 5 with the two-bootstraps-per-workgroup form of the wave-per-polynomial build in the automatic choice:
# variant tests (product and lab library), the headline with the choice on / off on this box, the profile of the bench command,
# the bench line, config 5 in its opt-in forms (wide launches)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_pbs.py tests/test_gpu_param_shapes.py -q -m gpu -x -k "variants or wave_per or generic_kernel_equals or automatic_choice or pbs_batch_bit_exact" > gpurun_out/r05_final_d_tests.txt 2>&1 || { tail -30 gpurun_out/r05_final_d_tests.txt; exit 1; }
tail -2 gpurun_out/r05_final_d_tests.txt
FHES_LAB=1 timeout -k 10 300 python -m pytest tests/test_gpu_pbs.py -q -m gpu -x -k "variants" > gpurun_out/r05_final_d_tests_lab.txt 2>&1 || { tail -30 gpurun_out/r05_final_d_tests_lab.txt; exit 1; }
tail -2 gpurun_out/r05_final_d_tests_lab.txt
O=gpurun_out/r05_probe34_wpp2_auto_ab.txt
: > $O
for A in 1 0 1 0; do
  echo "FHES_WPP_AUTO=$A" >> $O
  FHES_WPP_AUTO=$A timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"find_wall_ms_single": [0-9.]*\|"result_ok": [a-z]*' | tr '\n' ' ' >> $O
  echo >> $O
done
cat $O
echo "ab done" >> gpurun_out/heartbeat.txt
scripts/r05_profile.sh classic
echo "profile done" >> gpurun_out/heartbeat.txt
timeout -k 10 300 python bench.py > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_line.err || { tail -20 gpurun_out/r05_bench_line.err; exit 1; }
timeout -k 10 120 python scripts/config5.py 1 all logdepth > gpurun_out/r05_config5_g1_logdepth.jsonl 2>&1
python - <<'PY'
import json
l = json.loads([x for x in open("gpurun_out/r05_bench_line.json") if x.startswith("{")][-1])
r = l["roofline"]
print({k: l[k] for k in ("value", "ms_per_step", "finds_per_s", "find_wall_ms_single", "find_wall_ms_single_logdepth", "result_ok")},
      {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "frac", "traffic", "launch_ms", "pbs_per_launch")})
PY
cut -c1-160 gpurun_out/r05_config5_g1_logdepth.jsonl
