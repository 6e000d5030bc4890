#!/bin/bash
# round 5, probe 21: the split form as the default for small multi-bit launches -- the multi-bit tests, then bench lines and config 5
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_pbs.py tests/test_gpu_strings.py tests/test_gpu_fullsize_oracle.py::test_config5_find_512_multibit2_every_output_word_against_oracle "tests/test_gpu_config5_fullsize.py::test_config5_logdepth_form_every_word_against_oracle_and_every_block_against_the_restatement[rfind]" -x -q -m gpu > gpurun_out/r05_probe21_tests.txt 2>&1 || { tail -30 gpurun_out/r05_probe21_tests.txt; exit 1; }
tail -2 gpurun_out/r05_probe21_tests.txt
for P in multibit2 multibit3; do
  timeout -k 10 300 python bench.py --params $P --no-cpu-baseline > gpurun_out/r05_probe21_bench_$P.json 2>> gpurun_out/r05_probe21.err
done
python - <<'PY'
import json
for P in ("multibit2", "multibit3"):
    l = json.loads([x for x in open(f"gpurun_out/r05_probe21_bench_{P}.json") if x.startswith("{")][-1])
    print(P, {k: l[k] for k in ("value", "finds_per_s", "find_wall_ms_single", "find_wall_ms_single_logdepth", "result_ok")}, l["callers"]["find_wall_ms_8_callers"], [(c["op"][:24], c["wall_ms"], c["result_ok"]) for c in l["sharded"]["cases"]])
PY
for g in 2 3; do timeout -k 10 400 python scripts/config5.py $g all faithful > gpurun_out/r05_config5_g${g}_faithful.jsonl 2>&1; cut -c1-150 gpurun_out/r05_config5_g${g}_faithful.jsonl; done
