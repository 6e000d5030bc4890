#!/bin/bash
# PREPARED AT THE END OF ROUND 5 WITHOUT A DEVICE IN REACH (compiled, 252 registers, no spills; never run).
# Wave-per-polynomial build with FOUR bootstraps per workgroup (512 threads, one workgroup per CU, eight waves at both barriers):
#   lab variant 31: + the 1024 untwist factors in an LDS table of the workgroup (161 792 bytes of LDS) -- the 16 per-lane requests
#                   per CMUX that are left in the loop of variant 17 cost 0.3 % of the rate each (r05_probe28_wpp_ablations.txt)
#   lab variant 32: four per workgroup alone (separates the effect of the wider barrier from that of the table)
# Bits first (against the throughput build and the oracle, odd launch sizes: the last workgroup repeats the last bootstrap up to
# three times), then rates.  Gate for taking 31 into the automatic choice: >= 3 % above variant 17 at 4096 and 16 384 in the same run.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r06_probe1_wpp_four_per_workgroup.txt
FHES_LAB=1 python -m fhe_string_amd.build > gpurun_out/r06_probe1_build.txt 2>&1 || { tail -20 gpurun_out/r06_probe1_build.txt; exit 1; }
FHES_LAB=1 timeout -k 10 300 python - > $O 2>&1 <<'PY' || { tail -20 $O; exit 1; }
import sys
import numpy as np
sys.path.insert(0, ".")
from fhe_string_amd._lib import check
from tests.helpers import ptr, world
w = world("default")
ctx = w.ctx()
bad = 0
for count in (1, 7, 12, 1030):
    vals = (np.arange(count) * 5 % 16).astype(np.uint8)
    cts = w.encrypt(vals)
    t, poly = w.lut_poly(lambda x: (7 * x + 2) % 16)
    lid = check(w.lib.fhes_lut_register(ctx, ptr(t), 16))
    h_in = check(w.lib.fhes_arena_alloc(ctx, count)); h_out = check(w.lib.fhes_arena_alloc(ctx, count))
    check(w.lib.fhes_arena_upload(ctx, h_in, count, ptr(cts)))
    ref = np.stack([w.oracle_pbs(cts[i], poly) for i in (0, count - 1)])
    res = {}
    for variant in (3, 17, 32, 31):
        check(w.lib.fhes_arena_upload(ctx, h_out, count, ptr(np.zeros((count, w.W), dtype=np.uint64))))
        check(w.lib.fhes_set_kernel_variant(ctx, variant))
        check(w.lib.fhes_pbs_batch(ctx, count, ptr(np.arange(count + 1, dtype=np.int32)), ptr(np.arange(h_in, h_in + count, dtype=np.int32)), ptr(np.ones(count, dtype=np.int64)),
                                   ptr(np.zeros(count, dtype=np.uint64)), ptr(np.full(count, lid, dtype=np.int32)), ptr(np.arange(h_out, h_out + count, dtype=np.int32))))
        r = np.zeros((count, w.W), dtype=np.uint64)
        check(w.lib.fhes_arena_download(ctx, h_out, count, ptr(r)))
        res[variant] = r
        ok_o, ok_3 = bool(np.array_equal(r[[0, count - 1]], ref)), bool(np.array_equal(r, res[3]))
        bad += (not ok_o) + (not ok_3)
        print(f"count {count} variant {variant}: first / last equal the oracle: {ok_o}, all equal variant 3: {ok_3}", flush=True)
    check(w.lib.fhes_arena_free(ctx, h_in, count)); check(w.lib.fhes_arena_free(ctx, h_out, count))
check(w.lib.fhes_set_kernel_variant(ctx, 0))
print("BITS", "OK" if bad == 0 else f"WRONG ({bad})", flush=True)
sys.exit(1 if bad else 0)
PY
cat $O
FHES_LAB=1 timeout -k 10 500 python scripts/bench_pbs.py 3,17,32,31,17,32,31 1024,2048,4096,16384 default >> $O 2>&1
grep "variant=" $O
