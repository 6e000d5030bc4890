"""Lab: builds of the blind rotate against the oracle word for word, and against each other.
   python scripts/mb_lat_check.py multibit2 multibit3     the multi-bit builds (3, 0, 4, 16)
   python scripts/mb_lat_check.py default                 the classic builds (3, 0, 2, 17)"""
import sys

import numpy as np

sys.path.insert(0, ".")
from fhe_string_amd._lib import check
from tests.helpers import ptr, world

for which in sys.argv[1:] or ["multibit2", "multibit3"]:
    w = world(which)
    ctx = w.ctx()
    count = 12
    vals = (np.arange(count) * 5 % 16).astype(np.uint8)
    cts = w.encrypt(vals)
    t, poly = w.lut_poly(lambda x: (7 * x + 2) % 16)
    lid = check(w.lib.fhes_lut_register(ctx, ptr(t), 16))
    h_in = check(w.lib.fhes_arena_alloc(ctx, count))
    h_out = check(w.lib.fhes_arena_alloc(ctx, count))
    check(w.lib.fhes_arena_upload(ctx, h_in, count, ptr(cts)))
    ref = np.stack([w.oracle_pbs(cts[i], poly) for i in range(3)])
    res = {}
    for variant in ((3, 0, 2, 17) if w.p.grouping_factor <= 1 else (3, 0, 4, 16)):
        check(w.lib.fhes_set_kernel_variant(ctx, variant))
        tp = np.arange(count + 1, dtype=np.int32)
        ti = np.arange(h_in, h_in + count, dtype=np.int32)
        check(w.lib.fhes_pbs_batch(ctx, count, ptr(tp), ptr(ti), ptr(np.ones(count, dtype=np.int64)), ptr(np.zeros(count, dtype=np.uint64)),
                                   ptr(np.full(count, lid, dtype=np.int32)), ptr(np.arange(h_out, h_out + count, dtype=np.int32))))
        r = np.zeros((count, w.W), dtype=np.uint64)
        check(w.lib.fhes_arena_download(ctx, h_out, count, ptr(r)))
        res[variant] = r
        bad = int(np.sum(np.any(r[:3] != ref, axis=1)))
        dec_ok = list(w.decrypt(r)) == [(7 * int(v) + 2) % 16 for v in vals]
        print(f"{which} variant {variant}: {bad} of 3 ciphertexts differ from the oracle, all {count} equal variant 3: {bool(np.array_equal(r, res[3]))}, decrypt ok: {dec_ok}", flush=True)
    check(w.lib.fhes_set_kernel_variant(ctx, 0))
