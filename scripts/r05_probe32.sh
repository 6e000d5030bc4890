#!/bin/bash
# wave-per-polynomial build with TWO bootstraps per workgroup (lab variant 30): bits against the throughput build, then rates
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r05_probe32_wpp_two_per_workgroup.txt
FHES_LAB=1 timeout -k 10 200 python - > $O 2>&1 <<'PY' || { tail -20 $O; exit 1; }
import sys
import numpy as np
sys.path.insert(0, ".")
from fhe_string_amd._lib import check
from tests.helpers import ptr, world
w = world("default")
ctx = w.ctx()
for count in (7, 12):
    vals = (np.arange(count) * 5 % 16).astype(np.uint8)
    cts = w.encrypt(vals)
    t, poly = w.lut_poly(lambda x: (7 * x + 2) % 16)
    lid = check(w.lib.fhes_lut_register(ctx, ptr(t), 16))
    h_in = check(w.lib.fhes_arena_alloc(ctx, count)); h_out = check(w.lib.fhes_arena_alloc(ctx, count))
    check(w.lib.fhes_arena_upload(ctx, h_in, count, ptr(cts)))
    ref = np.stack([w.oracle_pbs(cts[i], poly) for i in (0, count - 1)])
    res = {}
    for variant in (3, 17, 30):
        check(w.lib.fhes_set_kernel_variant(ctx, variant))
        check(w.lib.fhes_pbs_batch(ctx, count, ptr(np.arange(count + 1, dtype=np.int32)), ptr(np.arange(h_in, h_in + count, dtype=np.int32)), ptr(np.ones(count, dtype=np.int64)),
                                   ptr(np.zeros(count, dtype=np.uint64)), ptr(np.full(count, lid, dtype=np.int32)), ptr(np.arange(h_out, h_out + count, dtype=np.int32))))
        r = np.zeros((count, w.W), dtype=np.uint64)
        check(w.lib.fhes_arena_download(ctx, h_out, count, ptr(r)))
        res[variant] = r
        print(f"count {count} variant {variant}: first / last equal the oracle: {bool(np.array_equal(r[[0, count - 1]], ref))}, all equal variant 3: {bool(np.array_equal(r, res[3]))}", flush=True)
check(w.lib.fhes_set_kernel_variant(ctx, 0))
PY
cat $O
FHES_LAB=1 timeout -k 10 400 python scripts/bench_pbs.py 3,17,30,17,30 512,1024,2048,4096,16384 default >> $O 2>&1
grep "variant=" $O
