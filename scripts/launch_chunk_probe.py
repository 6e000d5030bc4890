"""Developer probe (LAB build, FHES_LAB=1): does a launch boundary every few rounds keep the multi-bit g = 3 kernel out of its slow
state?  One process, one key, 16 384 bootstraps per call; the chunk size (FHES_LAUNCH_CHUNK, read by the lab build at every call)
alternates, so every setting sees the same process -- the slow state is a property of the process or of the launch, which of the two
is the question (profiles/r05_probe11_multibit3_process_counters.txt).  Same kernels on sub-ranges: bits cannot change;
the first / last 16 results are decrypted anyway.
    FHES_LAB=1 python scripts/launch_chunk_probe.py [multibit3] [passes]"""
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
    assert os.environ.get("FHES_LAB") == "1", "the knob exists in the lab build only"
    which = sys.argv[1] if len(sys.argv) > 1 else "multibit3"
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    lib = _lib.load(strict=False)
    p = Params()
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
    count, W = 16384, p.ct_words()
    vals = (np.arange(count) % 16).astype(np.uint8)
    cts = np.zeros((count, W), dtype=np.uint64)
    check(lib.fhes_client_encrypt_blocks(ck, ptr(vals), count, ptr(cts)))
    h_in = check(lib.fhes_arena_alloc(ctx, count))
    h_out = check(lib.fhes_arena_alloc(ctx, count))
    check(lib.fhes_arena_upload(ctx, h_in, count, ptr(cts)))
    t = np.array([(x + 1) % 16 for x in range(16)], dtype=np.uint64)
    lid = check(lib.fhes_lut_register(ctx, ptr(t), 16))
    keep = [np.arange(count + 1, dtype=np.int32), np.arange(h_in, h_in + count, dtype=np.int32), np.ones(count, dtype=np.int64),
            np.zeros(count, dtype=np.uint64), np.full(count, lid, dtype=np.int32), np.arange(h_out, h_out + count, dtype=np.int32)]
    args = (ctx, count) + tuple(ptr(a) for a in keep)
    check(lib.fhes_pbs_batch(*args))
    check(lib.fhes_sync(ctx))
    ref = None
    for ps in range(passes):
        for chunk in (16384, 4096, 2048, 1024, 512):
            os.environ["FHES_LAUNCH_CHUNK"] = str(chunk)
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
            res = np.zeros((count, W), dtype=np.uint64)
            check(lib.fhes_arena_download(ctx, h_out, count, ptr(res)))
            if ref is None:
                ref = res
            same = bool(np.array_equal(res, ref))
            ends = np.ascontiguousarray(np.concatenate([res[:16], res[-16:]]))
            dec = np.zeros(32, dtype=np.uint8)
            check(lib.fhes_client_decrypt_blocks(ck, ptr(ends), 32, ptr(dec)))
            ok = bool(np.array_equal(dec, (np.concatenate([vals[:16], vals[-16:]]) + 1) % 16))
            print(f"{which} pass {ps} chunk={chunk:6d} wall={dt*1e3:9.3f} ms  br={st.total_br_ms/reps:9.3f} ms  PBS/s={count/dt:10.1f}  "
                  f"bits_equal_first_call={same} decrypt_ok={ok}", flush=True)
    os.environ.pop("FHES_LAUNCH_CHUNK", None)


if __name__ == "__main__":
    main()
