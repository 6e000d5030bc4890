"""Developer tool: where do the cycles of one CMUX iteration go?  Runs the instrumented build of
blind-rotate variant 3 (variant 4, the default throughput kernel, has the same phases minus one barrier)."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from fhe_string_amd import _lib
from fhe_string_amd._lib import Params, check, check_ptr

NAMES = ["prefetch+rotate+digits+twist", "fwd pass A + write", "BARRIER 1", "read + fwd pass B + write", "BARRIER 2",
         "read + fwd pass C + DPP", "(barrier removed: spectra are written in place)", "spectrum write", "BARRIER 4", "read other + ext. product",
         "BARRIER 5", "DPP + inv pass C + write", "BARRIER 6", "read + inv pass B + write", "BARRIER 7",
         "read + inv pass A + untwist + round + acc", "BARRIER 8"]


def main():
    lib = _lib.load(strict=False)
    p = Params()
    check(lib.fhes_params_default(C.byref(p)))
    ck = check_ptr(lib.fhes_client_key_generate(C.byref(p), 255, 1))
    bsk = np.zeros(lib.fhes_bsk_words(C.byref(p)), dtype=np.uint64)
    ksk = np.zeros(lib.fhes_ksk_words(C.byref(p)), dtype=np.uint64)
    check(lib.fhes_client_key_make_server_keys(ck, bsk.ctypes.data, ksk.ctypes.data))
    ctx = check_ptr(lib.fhes_ctx_create(C.byref(p), 0))
    check(lib.fhes_upload_keys(ctx, bsk.ctypes.data, ksk.ctypes.data))
    ct = np.zeros((1, p.ct_words()), dtype=np.uint64)
    v = np.array([5], dtype=np.uint8)
    check(lib.fhes_client_encrypt_blocks(ck, v.ctypes.data, 1, ct.ctypes.data))
    h = check(lib.fhes_arena_alloc(ctx, 1))
    check(lib.fhes_arena_upload(ctx, h, 1, ct.ctypes.data))
    for count in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "512", "2048"])]:
        out = np.zeros(17)
        check(lib.fhes_debug_phase_profile(ctx, count, h, out.ctypes.data))
        tot = out.sum()
        print(f"--- {count} ciphertexts: {tot:.0f} cycles per CMUX iteration per wave")
        for n, c in zip(NAMES, out):
            print(f"   {n:45s} {c:8.0f}  {100 * c / tot:5.1f}%")
        print(f"   barriers total {out[[2,4,6,8,10,12,14,16]].sum():.0f} ({100*out[[2,4,6,8,10,12,14,16]].sum()/tot:.1f}%)")


if __name__ == "__main__":
    main()
