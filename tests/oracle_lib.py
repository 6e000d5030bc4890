"""ctypes loader of the CPU oracle (oracle/liboracle.so) -- test infrastructure only."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "liboracle.so")
_P = C.c_void_p
_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    lib = C.CDLL(PATH)
    sig = {
        "oracle_set_simd": (None, [C.c_int]),
        "oracle_get_simd": (C.c_int, []),
        "oracle_ctx_create": (_P, [_P]),
        "oracle_ctx_destroy": (None, [_P]),
        "oracle_set_keys": (C.c_int, [_P, _P, _P]),
        "oracle_bsk_fourier": (_P, [_P]),
        "oracle_twist_table": (_P, [_P]),
        "oracle_twiddle_table": (_P, [_P]),
        "oracle_fft_forward_i64": (None, [_P, _P, _P]),
        "oracle_fft_forward_internal": (None, [_P, _P, _P]),
        "oracle_psi_table": (_P, [_P]),
        "oracle_odd_fix_table": (_P, [_P]),
        "oracle_fft_inverse_torus": (None, [_P, _P, _P]),
        "oracle_negacyclic_mul_exact": (None, [_P, _P, _P, _P]),
        "oracle_make_lut_poly": (None, [_P, _P, _P]),
        "oracle_keyswitch": (None, [_P, _P, _P]),
        "oracle_keyswitch_batch": (None, [_P, _P, _P, C.c_int]),
        "oracle_modswitch": (None, [_P, _P, _P]),
        "oracle_blind_rotate": (None, [_P, _P, _P, _P]),
        "oracle_sample_extract": (None, [_P, _P, _P]),
        "oracle_blind_rotate_multibit": (None, [_P, _P, _P, _P]),
        "oracle_root_table": (_P, [_P]),
        "oracle_slot_exp_table": (_P, [_P]),
        "oracle_pbs": (None, [_P, _P, _P, _P]),
        "oracle_exec_levels": (C.c_int, [_P, _P, C.c_size_t, C.c_int32] + [_P] * 8 + [C.c_int32]),
        "oracle_lwe_phase": (C.c_uint64, [_P, _P, C.c_int32]),
        "oracle_keygen": (C.c_int, [_P, C.c_uint64, _P, _P, _P, _P]),
        "oracle_encrypt_blocks": (None, [_P, _P, C.c_uint64, _P, C.c_int64, _P]),
        "oracle_decrypt_blocks": (None, [_P, _P, _P, C.c_int64, _P]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
