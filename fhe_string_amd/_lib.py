"""ctypes binding of the C ABI (include/fhe_string_amd.h).  No torch types cross it."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FHES_LIB: developer knob -- another build of the same library (compiler-flag A/B runs, scripts/flag_ab.sh)
LIB_PATH = os.environ.get("FHES_LIB") or os.path.join(_HERE, "libfhe_string_amd_lab.so" if os.environ.get("FHES_LAB") == "1" else "libfhe_string_amd.so")


class Params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "lwe_dim", "glwe_dim", "poly_size", "pbs_base_log", "pbs_level", "ks_base_log",
        "ks_level", "message_bits", "carry_bits", "grouping_factor")] + [
        ("lwe_noise_std", C.c_double), ("glwe_noise_std", C.c_double)]

    def ct_words(self):
        return self.glwe_dim * self.poly_size + 1


class Stats(C.Structure):
    _fields_ = [("pbs_count", C.c_uint64), ("pbs_launches", C.c_uint64), ("linear_count", C.c_uint64),
                ("arena_capacity", C.c_uint64), ("arena_in_use", C.c_uint64),
                ("last_ks_ms", C.c_double), ("last_br_ms", C.c_double),
                ("total_ks_ms", C.c_double), ("total_br_ms", C.c_double),
                ("timed_launches", C.c_uint64), ("host_bytes_up", C.c_uint64), ("host_bytes_down", C.c_uint64)]


class FheError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[fhes error {code}] {msg}")
        self.code = code


_P = C.c_void_p
_PP = C.POINTER(Params)
_i32p = C.POINTER(C.c_int32)

# name -> (restype, argtypes)
SIGNATURES = {
    "fhes_params_default": (C.c_int, [_PP]),
    "fhes_params_toy": (C.c_int, [_PP]),
    "fhes_params_multibit": (C.c_int, [_PP, C.c_int32]),
    "fhes_params_toy_multibit": (C.c_int, [_PP, C.c_int32]),
    "fhes_ct_words": (C.c_int64, [_PP]),
    "fhes_bsk_words": (C.c_int64, [_PP]),
    "fhes_ksk_words": (C.c_int64, [_PP]),
    "fhes_last_error": (C.c_char_p, []),
    "fhes_client_key_generate": (_P, [_PP, C.c_uint64, C.c_uint64]),
    "fhes_client_key_destroy": (None, [_P]),
    "fhes_client_key_params": (C.c_int, [_P, _PP]),
    "fhes_client_key_num_blocks_char": (C.c_int32, [_P]),
    "fhes_client_key_num_blocks_usize": (C.c_int32, [_P]),
    "fhes_client_key_big_sk": (_P, [_P]),
    "fhes_client_key_small_sk": (_P, [_P]),
    "fhes_client_key_make_server_keys": (C.c_int, [_P, _P, _P]),
    "fhes_client_encrypt_blocks": (C.c_int, [_P, _P, C.c_int64, _P]),
    "fhes_client_decrypt_blocks": (C.c_int, [_P, _P, C.c_int64, _P]),
    "fhes_ctx_create": (_P, [_PP, C.c_int32]),
    "fhes_ctx_destroy": (None, [_P]),
    "fhes_upload_keys": (C.c_int, [_P, _P, _P]),
    "fhes_lut_register": (C.c_int32, [_P, _P, C.c_int32]),
    "fhes_arena_alloc": (C.c_int32, [_P, C.c_int32]),
    "fhes_arena_free": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "fhes_arena_upload": (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    "fhes_arena_download": (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    "fhes_pbs_batch": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "fhes_linear_batch": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P]),
    "fhes_sync": (C.c_int, [_P]),
    "fhes_get_stats": (C.c_int, [_P, C.POINTER(Stats)]),
    "fhes_reset_stats": (C.c_int, [_P]),
    "fhes_set_timing": (C.c_int, [_P, C.c_int32]),
    "fhes_set_kernel_variant": (C.c_int, [_P, C.c_int32]),
    "fhes_debug_bsk_fourier": (C.c_int, [_P, _P]),
    "fhes_debug_fft_tables": (C.c_int, [_PP, _P, _P]),
    "fhes_debug_fwd_tables": (C.c_int, [_P, _P, _P]),
    "fhes_debug_keyswitch": (C.c_int, [_P, C.c_int32, _P, _P]),
    "fhes_debug_keyswitch_words": (C.c_int, [_P, C.c_int32, _P, _P]),
    "fhes_set_keyswitch_mfma_min": (C.c_int, [_P, C.c_int32]),
    "fhes_program_compile": (_P, [_PP, C.c_char_p, _P, C.c_int32, C.c_int32]),
    "fhes_program_destroy": (None, [_P]),
    "fhes_program_scratch_cells": (C.c_int32, [_P]),
    "fhes_program_out_meta": (C.c_int, [_P, _P, _P]),
    "fhes_program_info": (C.c_int, [_P, _P]),
    "fhes_program_out_parts": (C.c_int32, [_P, _P, C.c_int32]),
    "fhes_op_signature": (C.c_char_p, [C.c_char_p]),
    "fhes_program_export": (C.c_int, [_P] + [_P] * 10),
    "fhes_program_run": (C.c_int, [_P, _P, _P, _P]),
    "fhes_program_run_batch": (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    "fhes_server_key_create": (_P, [_P, C.c_int32]),
    "fhes_server_key_destroy": (None, [_P]),
    "fhes_server_key_ctx": (_P, [_P]),
    "fhes_server_key_info": (C.c_int, [_P, _P, _P, _P]),
    "fhes_server_key_set_logdepth": (C.c_int, [_P, C.c_int32]),
    "fhes_string_encrypt": (_P, [_P, C.c_char_p, C.c_int64, _i32p]),
    "fhes_string_trivial": (_P, [_P, C.c_char_p, _i32p]),
    "fhes_usize_encrypt": (_P, [_P, C.c_uint64]),
    "fhes_usize_trivial": (_P, [_P, C.c_uint64]),
    "fhes_radix_encrypt": (_P, [_P, C.c_uint64]),
    "fhes_radix_decrypt": (C.c_int, [_P, _P, _P]),
    "fhes_ct_destroy": (None, [_P]),
    "fhes_ct_num_blocks": (C.c_int32, [_P]),
    "fhes_string_decrypt": (C.c_int, [_P, _P, _P, C.c_int64]),
    "fhes_usize_decrypt": (C.c_int, [_P, _P, C.POINTER(C.c_uint64)]),
    "fhes_bool_decrypt": (C.c_int, [_P, _P, _i32p]),
    "fhes_ct_decrypt_blocks": (C.c_int, [_P, _P, _P]),
    "fhes_client_key_serialize": (C.c_int64, [_P, _P, C.c_int64]),
    "fhes_client_key_deserialize": (_P, [_P, C.c_int64]),
    "fhes_ct_serialize": (C.c_int64, [_P, _P, C.c_int64]),
    "fhes_ct_deserialize": (_P, [_P, C.c_int64]),
    "fhes_ct_block_meta": (C.c_int, [_P, _P, _P]),
    "fhes_ct_slice": (_P, [_P, _P, C.c_int32, C.c_int32]),
    "fhes_ct_concat": (_P, [_P, _P, C.c_int32]),
    "fhes_server_key_serialize": (C.c_int64, [_P, _P, C.c_int64]),
    "fhes_client_key_serialize_server_key": (C.c_int64, [_P, _P, C.c_int64]),
    "fhes_ct_clone": (_P, [_P]),
    "fhes_download_keys": (C.c_int, [_P, _P, _P]),
    "fhes_server_key_deserialize": (_P, [_P, C.c_int64, C.c_int32]),
    "fhes_string_op": (C.c_int, [_P, C.c_char_p, _P, C.c_int32, _P, C.c_int32, _P, C.c_int32, _i32p]),
    # multi-GPU (one process per GPU)
    "fhes_comm_unique_id": (C.c_int, [_P]),
    "fhes_comm_create_rccl": (_P, [_P, C.c_int32, C.c_int32, _P]),
    "fhes_comm_create_host": (_P, [_P, C.c_int32, C.c_int32, _P, _P, _P]),
    "fhes_comm_destroy": (None, [_P]),
    "fhes_comm_self_test": (C.c_int, [_P]),
    "fhes_rccl_library_path": (C.c_char_p, []),
    "fhes_rccl_hip_runtime_path": (C.c_char_p, []),
    "fhes_comm_rank": (C.c_int32, [_P]),
    "fhes_comm_world": (C.c_int32, [_P]),
    "fhes_comm_is_rccl": (C.c_int32, [_P]),
    "fhes_shard_range": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "fhes_comm_all_gather_host": (C.c_int, [_P, _P, _P, C.c_int64]),
    "fhes_comm_all_gather_cts": (C.c_int, [_P, C.c_int32, C.c_int32]),
    "fhes_comm_broadcast_cts": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32]),
    "fhes_comm_broadcast_keys": (C.c_int, [_P, C.c_int32]),
    "fhes_sharded_search": (C.c_int, [_P, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i32p, C.c_int32, _i32p]),
    "fhes_program_depth": (C.c_int32, [_P]),
    "fhes_sharded_op": (C.c_int, [_P, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i32p, C.c_int32, _i32p]),
    "fhes_sharded_op_ex": (C.c_int, [_P, C.c_char_p, _i32p, _i32p, C.c_int32, C.POINTER(C.c_int64), C.c_int32, C.c_int32, _i32p, _i32p, C.c_int32, _i32p]),
    "fhes_comm_create_host_exec": (_P, [_P, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "fhes_server_key_set_comm": (C.c_int, [_P, _P]),
}

ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
BCAST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32)
# fhes_executor (include/fhe_string_amd.h): the planner of fhes_sharded_op over host-memory rows
EXEC_ALLOC_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32)
EXEC_RELEASE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32)
EXEC_ROWS_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int32)
EXEC_RUN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32))


class Executor(C.Structure):
    _fields_ = [("user", C.c_void_p), ("row_words", C.c_int64), ("alloc", EXEC_ALLOC_FN), ("release", EXEC_RELEASE_FN),
                ("rows", EXEC_ROWS_FN), ("run", EXEC_RUN_FN)]

_lib = None


def load(strict=True):
    """Loads the product library.  There is no fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FheError(-2, f"{LIB_PATH} not built; run `python -m fhe_string_amd.build` "
                           "(the HIP extension is required, there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if strict:
                raise
            continue
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc is None or (isinstance(rc, int) and rc < 0):
        msg = load().fhes_last_error()
        raise FheError(rc if rc is not None else -1, msg.decode() if msg else "unknown error")
    return rc


def check_ptr(p):
    if not p:
        msg = load().fhes_last_error()
        raise FheError(-1, msg.decode() if msg else "unknown error")
    return p
