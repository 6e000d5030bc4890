"""Lab (FHES_LAB=1, FHES_BSK_E_CANDIDATES=K): the SAME multi-bit key at K addresses in one process -- does the rate of the wide
blind rotate follow the placement of the key?"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from fhe_string_amd import _lib
from fhe_string_amd._lib import Params, Stats, check, check_ptr


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "multibit3"
    K = int(os.environ.get("FHES_BSK_E_CANDIDATES", "1"))
    lib = _lib.load(strict=False)
    p = Params()
    check(lib.fhes_params_multibit(C.byref(p), int(which[-1])))
    ck = check_ptr(lib.fhes_client_key_generate(C.byref(p), 255, 1))
    bsk = np.zeros(lib.fhes_bsk_words(C.byref(p)), dtype=np.uint64)
    ksk = np.zeros(lib.fhes_ksk_words(C.byref(p)), dtype=np.uint64)
    check(lib.fhes_client_key_make_server_keys(ck, ptr(bsk), ptr(ksk)))
    ctx = check_ptr(lib.fhes_ctx_create(C.byref(p), 0))
    check(lib.fhes_upload_keys(ctx, ptr(bsk), ptr(ksk)))
    count = 16384
    W = p.ct_words()
    vals = (np.arange(count) % 16).astype(np.uint8)
    cts = np.zeros((count, W), dtype=np.uint64)
    check(lib.fhes_client_encrypt_blocks(ck, ptr(vals), count, ptr(cts)))
    h_in = check(lib.fhes_arena_alloc(ctx, count))
    h_out = check(lib.fhes_arena_alloc(ctx, count))
    check(lib.fhes_arena_upload(ctx, h_in, count, ptr(cts)))
    t = np.array([(x + 1) % 16 for x in range(16)], dtype=np.uint64)
    lid = check(lib.fhes_lut_register(ctx, ptr(t), 16))
    term_ptr = np.arange(count + 1, dtype=np.int32)
    term_idx = np.arange(h_in, h_in + count, dtype=np.int32)
    term_coef = np.ones(count, dtype=np.int64)
    consts = np.zeros(count, dtype=np.uint64)
    lut_ids = np.full(count, lid, dtype=np.int32)
    outs = np.arange(h_out, h_out + count, dtype=np.int32)
    args = (ctx, count, ptr(term_ptr), ptr(term_idx), ptr(term_coef), ptr(consts), ptr(lut_ids), ptr(outs))
    for rnd in range(2):
        for i in range(K):
            check(lib.fhes_set_kernel_variant(ctx, 100 + i))
            check(lib.fhes_pbs_batch(*args))
            check(lib.fhes_sync(ctx))
            t0 = time.time()
            for _ in range(3):
                check(lib.fhes_pbs_batch(*args))
            check(lib.fhes_sync(ctx))
            dt = (time.time() - t0) / 3
            print(f"{which} round {rnd} key copy {i}: {dt * 1e3:8.2f} ms  {count / dt:9.0f} PBS/s", flush=True)


if __name__ == "__main__":
    main()
