#!/bin/bash
# round 4: the records of the final build -- full GPU suite, smoke, bench lines (classic / multibit2 / multibit3)
mkdir -p gpurun_out
( time python -m pytest tests -q -m gpu --durations=15 ) > gpurun_out/r04_gpu_tests.txt 2>&1
tail -25 gpurun_out/r04_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.txt 2>&1; tail -2 gpurun_out/r04_smoke.txt
python bench.py --steps 5 --warmup 1 > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench_line.err; tail -c 600 gpurun_out/r04_bench_line.json; echo
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --params multibit2 > gpurun_out/r04_bench_line_multibit2.json 2>/dev/null
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --params multibit3 > gpurun_out/r04_bench_line_multibit3.json 2>/dev/null
for f in gpurun_out/r04_bench_line*.json; do python -c "import sys,json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline'].get('achieved'), d['roofline'].get('frac'))"; done
