"""Lab: runs the stamped latency build (variant 26) on a few bootstraps and prints, per wave of workgroup 0, the s_memtime
stamps of one CMUX iteration relative to the earliest one (cycles).  FHES_LAT_STAMPS names the dump file."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, ".")
path = os.environ.setdefault("FHES_LAT_STAMPS", "gpurun_out/lat_stamps.bin")
from fhe_string_amd import _lib
from fhe_string_amd._lib import Params, check, check_ptr


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


NAMES = ["top", "digits done", "pass A done", "X1 stored: at B1", "B1 released", "B,perm,C done", "X2 round trip issued", "D,perm,E,dpp done",
         "S stored: at B3", "B3 released", "MAC done", "dpp,E',perm,D' done", "X2' round trip issued", "C',perm,B' done", "X1' stored: at B5",
         "B5 released", "A' done", "rounded, image stored: at B6", "B6 released"]


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    variant = int(sys.argv[2]) if len(sys.argv) > 2 else 26
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
    check(lib.fhes_set_kernel_variant(ctx, variant))
    term_ptr = np.arange(count + 1, dtype=np.int32)
    term_idx = np.arange(h_in, h_in + count, dtype=np.int32)
    term_coef = np.ones(count, dtype=np.int64)
    consts = np.zeros(count, dtype=np.uint64)
    lut_ids = np.full(count, lid, dtype=np.int32)
    outs = np.arange(h_out, h_out + count, dtype=np.int32)
    for _ in range(3):
        check(lib.fhes_pbs_batch(ctx, count, ptr(term_ptr), ptr(term_idx), ptr(term_coef), ptr(consts), ptr(lut_ids), ptr(outs)))
    check(lib.fhes_sync(ctx))
    lib.fhes_ctx_destroy(ctx)
    st = np.fromfile(path, dtype=np.uint64).reshape(256, 8, 32)[:, :, :len(NAMES)].astype(np.int64)
    for wg in range(min(count, 2)):
        s = st[wg]
        t0 = s[:, 0].min()
        print(f"workgroup {wg}: one CMUX iteration = {int(s[:, 18].max() - t0)} cycles (max over waves); stamps relative to the earliest wave's top")
        print("%-34s" % "point" + "".join(f"  w{w}(h{w >> 2})" for w in range(8)) + "    delta(max)")
        prev = None
        for k, name in enumerate(NAMES):
            row = s[:, k] - t0
            d = "" if prev is None else f"{int(row.max() - prev.max()):8d}"
            print("%-34s" % name + "".join(f"{int(v):8d}" for v in row) + "    " + d)
            prev = row


if __name__ == "__main__":
    main()
