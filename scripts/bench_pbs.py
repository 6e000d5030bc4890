"""Developer micro-benchmark: raw fhes_pbs_batch throughput by batch size (no oracle involved)."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from fhe_string_amd import _lib
from fhe_string_amd._lib import Params, Stats, check, check_ptr


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def main():
    lib = _lib.load(strict=False)
    p = Params()
    which = sys.argv[3] if len(sys.argv) > 3 else "default"   # default | multibit2 | multibit3
    if which.startswith("multibit"):
        check(lib.fhes_params_multibit(C.byref(p), int(which[-1])))
    else:
        check(lib.fhes_params_default(C.byref(p)))
    ck = check_ptr(lib.fhes_client_key_generate(C.byref(p), 255, 1))
    bsk = np.zeros(lib.fhes_bsk_words(C.byref(p)), dtype=np.uint64)
    ksk = np.zeros(lib.fhes_ksk_words(C.byref(p)), dtype=np.uint64)
    check(lib.fhes_client_key_make_server_keys(ck, ptr(bsk), ptr(ksk)))
    ctx = check_ptr(lib.fhes_ctx_create(C.byref(p), 0))
    check(lib.fhes_upload_keys(ctx, ptr(bsk), ptr(ksk)))
    variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
    sizes = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 8, 64, 256, 512, 1024, 4096]
    W = p.ct_words()
    maxc = max(sizes)
    vals = (np.arange(maxc) % 16).astype(np.uint8)
    cts = np.zeros((maxc, W), dtype=np.uint64)
    check(lib.fhes_client_encrypt_blocks(ck, ptr(vals), maxc, ptr(cts)))
    h_in = check(lib.fhes_arena_alloc(ctx, maxc))
    h_out = check(lib.fhes_arena_alloc(ctx, maxc))
    check(lib.fhes_arena_upload(ctx, h_in, maxc, ptr(cts)))
    t = np.array([(x + 1) % 16 for x in range(16)], dtype=np.uint64)
    lid = check(lib.fhes_lut_register(ctx, ptr(t), 16))
    for variant in variants:
        check(lib.fhes_set_kernel_variant(ctx, variant))
        for count in sizes:
            term_ptr = np.arange(count + 1, dtype=np.int32)
            term_idx = np.arange(h_in, h_in + count, dtype=np.int32)
            term_coef = np.ones(count, dtype=np.int64)
            consts = np.zeros(count, dtype=np.uint64)
            lut_ids = np.full(count, lid, dtype=np.int32)
            outs = np.arange(h_out, h_out + count, dtype=np.int32)
            args = (ctx, count, ptr(term_ptr), ptr(term_idx), ptr(term_coef), ptr(consts), ptr(lut_ids), ptr(outs))
            check(lib.fhes_pbs_batch(*args))
            check(lib.fhes_sync(ctx))
            check(lib.fhes_set_timing(ctx, 1))
            check(lib.fhes_reset_stats(ctx))
            reps = 3
            t0 = time.time()
            for _ in range(reps):
                check(lib.fhes_pbs_batch(*args))
            check(lib.fhes_sync(ctx))
            dt = (time.time() - t0) / reps
            st = Stats()
            check(lib.fhes_get_stats(ctx, C.byref(st)))
            check(lib.fhes_set_timing(ctx, 0))
            res = np.zeros((min(count, 16), W), dtype=np.uint64)
            check(lib.fhes_arena_download(ctx, h_out, min(count, 16), ptr(res)))
            dec = np.zeros(len(res), dtype=np.uint8)
            check(lib.fhes_client_decrypt_blocks(ck, ptr(res), len(res), ptr(dec)))
            ok = bool(np.array_equal(dec, (vals[:len(res)] + 1) % 16))
            print(f"{which} variant={variant} count={count:6d} wall={dt*1e3:9.3f} ms  ks={st.total_ks_ms/reps:8.3f} ms "
                  f"br={st.total_br_ms/reps:9.3f} ms  PBS/s={count/dt:10.1f}  decrypt_ok={ok}", flush=True)


if __name__ == "__main__":
    main()
