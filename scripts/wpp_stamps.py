"""Lab: runs the stamped wave-per-polynomial build (variant 27; FHES_LAB=1 build) and prints the s_memtime stamps of ONE CMUX
iteration: per phase the median over all stamped waves of (stamp[k+1] - stamp[k]), for a launch of `count` bootstraps.
   FHES_LAB=1 python scripts/wpp_stamps.py 256 1024"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, ".")
path = os.environ.setdefault("FHES_LAT_STAMPS", "gpurun_out/wpp_stamps.bin")
from fhe_string_amd import _lib
from fhe_string_amd._lib import Params, check, check_ptr


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


NAMES = ["digits", "pass 1", "exchange 1 (16 stores, 16 loads issued) + twiddle requests", "pass 2", "exchange 2 + row requests",
         "pass 3, hand-over stores", "row requests, wait at B1", "external product", "inverse bits 0, 1", "wait at B2",
         "exchange 2'", "pass 2'", "exchange 1'", "pass 1'", "untwist, rounding, image"]


def one(count):
    lib = _lib.load(strict=False)
    p = Params()
    check(lib.fhes_params_default(C.byref(p)))
    ck = check_ptr(lib.fhes_client_key_generate(C.byref(p), 255, 1))
    bsk = np.zeros(lib.fhes_bsk_words(C.byref(p)), dtype=np.uint64)
    ksk = np.zeros(lib.fhes_ksk_words(C.byref(p)), dtype=np.uint64)
    check(lib.fhes_client_key_make_server_keys(ck, ptr(bsk), ptr(ksk)))
    ctx = check_ptr(lib.fhes_ctx_create(C.byref(p), 0))
    check(lib.fhes_upload_keys(ctx, ptr(bsk), ptr(ksk)))
    W = p.ct_words()
    vals = (np.arange(count) % 16).astype(np.uint8)
    cts = np.zeros((count, W), dtype=np.uint64)
    check(lib.fhes_client_encrypt_blocks(ck, ptr(vals), count, ptr(cts)))
    h_in = check(lib.fhes_arena_alloc(ctx, count))
    h_out = check(lib.fhes_arena_alloc(ctx, count))
    check(lib.fhes_arena_upload(ctx, h_in, count, ptr(cts)))
    t = np.array([(x + 1) % 16 for x in range(16)], dtype=np.uint64)
    lid = check(lib.fhes_lut_register(ctx, ptr(t), 16))
    check(lib.fhes_set_kernel_variant(ctx, 27))
    args = (ptr(np.arange(count + 1, dtype=np.int32)), ptr(np.arange(h_in, h_in + count, dtype=np.int32)), ptr(np.ones(count, dtype=np.int64)),
            ptr(np.zeros(count, dtype=np.uint64)), ptr(np.full(count, lid, dtype=np.int32)), ptr(np.arange(h_out, h_out + count, dtype=np.int32)))
    for _ in range(2):
        check(lib.fhes_pbs_batch(ctx, count, *args))
    check(lib.fhes_sync(ctx))
    lib.fhes_ctx_destroy(ctx)
    st = np.fromfile(path, dtype=np.uint64).reshape(2048, 32)[:2 * min(count, 1024), :16].astype(np.int64)
    dt = np.diff(st, axis=1)
    total = st[:, 15] - st[:, 0]
    print(f"launch of {count} bootstraps: one CMUX iteration, stamp 0 -> 15: median {int(np.median(total))} cycles (min {int(total.min())}, max {int(total.max())}) over {len(st)} waves")
    for k, name in enumerate(NAMES):
        print(f"  {k:2d} -> {k + 1:2d}  median {int(np.median(dt[:, k])):6d}   p10 {int(np.percentile(dt[:, k], 10)):6d}   p90 {int(np.percentile(dt[:, k], 90)):6d}   {name}")
    sys.stdout.flush()


if __name__ == "__main__":
    for c in [int(v) for v in sys.argv[1:]] or [256, 1024]:
        one(c)
