"""Developer micro-benchmark: raw fhes_pbs_batch throughput of the GENERIC blind rotate on parameter shapes other than
2_2, at realistic key sizes (values as recalled from tfhe-rs 0.5; noise parameters irrelevant for timing), and of the
generic kernel on the default set (variant 12) next to variant 5."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from fhe_string_amd import _lib
from fhe_string_amd._lib import Params, check, check_ptr


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


SETS = {
    "1_1 (k=3, N=512, n=702, 2^18 x 1)": dict(message_bits=1, carry_bits=1, glwe_dim=3, poly_size=512, pbs_level=1, pbs_base_log=18, lwe_dim=702, ks_level=4, ks_base_log=3),
    "3_3 (k=1, N=8192, n=864, 2^15 x 2)": dict(message_bits=3, carry_bits=3, glwe_dim=1, poly_size=8192, pbs_level=2, pbs_base_log=15, lwe_dim=864, ks_level=6, ks_base_log=3),
    "k=2, N=1024, n=742, 2^12 x 2": dict(message_bits=2, carry_bits=2, glwe_dim=2, poly_size=1024, pbs_level=2, pbs_base_log=12, lwe_dim=742),
}


def run(lib, p, label, count, variant=0):
    ck = check_ptr(lib.fhes_client_key_generate(C.byref(p), 255, 1))
    bsk = np.zeros(lib.fhes_bsk_words(C.byref(p)), dtype=np.uint64)
    ksk = np.zeros(lib.fhes_ksk_words(C.byref(p)), dtype=np.uint64)
    check(lib.fhes_client_key_make_server_keys(ck, ptr(bsk), ptr(ksk)))
    ctx = check_ptr(lib.fhes_ctx_create(C.byref(p), 0))
    check(lib.fhes_upload_keys(ctx, ptr(bsk), ptr(ksk)))
    check(lib.fhes_set_kernel_variant(ctx, variant))
    W = p.ct_words()
    space = 1 << (p.message_bits + p.carry_bits)
    vals = (np.arange(count) % space).astype(np.uint8)
    cts = np.zeros((count, W), dtype=np.uint64)
    check(lib.fhes_client_encrypt_blocks(ck, ptr(vals), count, ptr(cts)))
    h_in = check(lib.fhes_arena_alloc(ctx, count))
    h_out = check(lib.fhes_arena_alloc(ctx, count))
    check(lib.fhes_arena_upload(ctx, h_in, count, ptr(cts)))
    t = np.array([(x + 1) % space for x in range(space)], dtype=np.uint64)
    lid = check(lib.fhes_lut_register(ctx, ptr(t), space))
    args = (ctx, count, ptr(np.arange(count + 1, dtype=np.int32)), ptr(np.arange(h_in, h_in + count, dtype=np.int32)),
            ptr(np.ones(count, dtype=np.int64)), ptr(np.zeros(count, dtype=np.uint64)), ptr(np.full(count, lid, dtype=np.int32)),
            ptr(np.arange(h_out, h_out + count, dtype=np.int32)))
    check(lib.fhes_pbs_batch(*args))
    check(lib.fhes_sync(ctx))
    t0 = time.time()
    check(lib.fhes_pbs_batch(*args))
    check(lib.fhes_sync(ctx))
    dt = time.time() - t0
    res = np.zeros((min(count, 8), W), dtype=np.uint64)
    check(lib.fhes_arena_download(ctx, h_out, len(res), ptr(res)))
    dec = np.zeros(len(res), dtype=np.uint8)
    check(lib.fhes_client_decrypt_blocks(ck, ptr(res), len(res), ptr(dec)))
    print(f"{label:42s} count={count:5d} wall={dt * 1e3:9.2f} ms  PBS/s={count / dt:9.1f}  first results {list(dec[:4])}", flush=True)
    lib.fhes_ctx_destroy(ctx)
    lib.fhes_client_key_destroy(ck)


def main():
    lib = _lib.load(strict=False)
    for label, sh in SETS.items():
        p = Params()
        check(lib.fhes_params_default(C.byref(p)))
        for k, v in sh.items():
            setattr(p, k, v)
        p.lwe_noise_std, p.glwe_noise_std = 2.0 ** -40, 2.0 ** -60
        for count in (256, 2048):
            run(lib, p, label, count)
    p = Params()
    check(lib.fhes_params_default(C.byref(p)))
    for variant, name in ((5, "default set, variant 5"), (12, "default set, generic kernel")):
        run(lib, p, name, 2048, variant)


if __name__ == "__main__":
    main()
