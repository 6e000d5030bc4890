#!/bin/bash
# round 4 GPU probe 9: the multi-rank bench path end to end on ONE GPU (2 ranks on device 0, gloo instead of RCCL: a rehearsal of the
# launcher contract, the barrier / max-over-ranks timing, the sharded section and the exit-code exchange -- not a scaling result)
mkdir -p gpurun_out
FHES_BENCH_REHEARSAL=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r04_bench_rehearsal_2ranks.json 2> gpurun_out/r04_bench_rehearsal_2ranks.err
echo "rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_bench_rehearsal_2ranks.json").read().strip().splitlines()[-1])
print(d["n_gpus"], d["value"], d["ms_per_step"], d.get("rehearsal"), d["sharded"].get("transport"), [ (c["op"][:30], c["wall_ms"], c["result_ok"]) for c in d["sharded"]["cases"]])
PY
