"""Host-side sanitizer run.  tests/asan builds csrc/client.cpp,
csrc/objects.cpp and csrc/circuit/*.cpp with `g++ -fsanitize=address,undefined` into one executable (device
entry points = the no-device answers of tests/asan/device_stubs.cpp) and drives the circuit compiler, the client
and the (de)serialisers, including ~8000 truncated / corrupted blobs.  Pass = exit code 0 and no sanitizer report.
Round 3's first run of it found four defects (32-bit overflow of k * N and of level * base_log on hostile
parameter blocks, memcpy into an empty vector for a zero-block ciphertext blob, and an out-of-bounds read when a
ciphertext blob's width differs from the key's): each is now an FHES_ERR_INVALID."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from fhe_string_amd._lib import FheError, Params, check, check_ptr
from tests.helpers import ptr

HERE = os.path.dirname(os.path.abspath(__file__))


def test_sanitizer_harness_is_clean():
    d = os.path.join(HERE, "asan")
    subprocess.check_call(["make", "-C", d, "asan"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([os.path.join(d, "_build", "harness_asan")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-6000:]
    assert "ok (0 check(s) failed)" in r.stdout


def _blobs(lib):
    p = Params()
    check(lib.fhes_params_toy(C.byref(p)))
    ck = check_ptr(lib.fhes_client_key_generate(C.byref(p), 255, 7))
    n = lib.fhes_client_key_serialize(ck, None, 0)
    ck_blob = np.zeros(n, dtype=np.uint8)
    assert lib.fhes_client_key_serialize(ck, ptr(ck_blob), n) == n
    st = C.c_int32(0)
    s = check_ptr(lib.fhes_string_encrypt(ck, b"abc", 5, C.byref(st)))
    m = lib.fhes_ct_serialize(s, None, 0)
    ct_blob = np.zeros(m, dtype=np.uint8)
    assert lib.fhes_ct_serialize(s, ptr(ct_blob), m) == m
    return p, ck, ck_blob, s, ct_blob


def test_malformed_blobs_are_invalid_through_the_shipped_library(libs):
    """the same refusals from libfhe_string_amd.so itself (no sanitizer here: status codes and messages)"""
    lib, _ = libs
    p, ck, ck_blob, s, ct_blob = _blobs(lib)
    for blob, parse in ((ck_blob, lib.fhes_client_key_deserialize), (ct_blob, lib.fhes_ct_deserialize)):
        for cut in (0, 3, 11, 23, len(blob) // 2, len(blob) - 1):
            b = np.ascontiguousarray(blob[:cut])
            assert not parse(ptr(b) if cut else None, cut), cut
            assert lib.fhes_last_error()
        bad = blob.copy()
        bad[0] ^= 0xFF                                   # magic
        assert not parse(ptr(bad), len(bad))
        bad = blob.copy()
        bad[8] = 9                                       # kind
        assert not parse(ptr(bad), len(bad))
    # wrong kind: a ciphertext blob is not a key and vice versa
    assert not lib.fhes_client_key_deserialize(ptr(ct_blob), len(ct_blob))
    assert not lib.fhes_ct_deserialize(ptr(ck_blob), len(ck_blob))
    assert not lib.fhes_server_key_deserialize(ptr(ck_blob), len(ck_blob), 0)
    # oversized counts in the header
    bad = ct_blob.copy()
    bad[12:16] = np.frombuffer(np.int32(2 ** 30).tobytes(), dtype=np.uint8)      # n_blocks
    assert not lib.fhes_ct_deserialize(ptr(bad), len(bad))
    bad = ct_blob.copy()
    bad[16:24] = np.frombuffer(np.int64(2 ** 40).tobytes(), dtype=np.uint8)      # ct_words
    assert not lib.fhes_ct_deserialize(ptr(bad), len(bad))
    bad = ct_blob.copy()
    bad[12:16] = 0                                                                # zero blocks
    assert not lib.fhes_ct_deserialize(ptr(bad), len(bad))
    bad = ct_blob.copy()
    bad[24] = 200                                                                 # degree byte of block 0
    assert not lib.fhes_ct_deserialize(ptr(bad), len(bad))
    # hostile parameter block inside a key blob: glwe_dim * poly_size would overflow 32 bits
    bad = ck_blob.copy()
    off = 12 + Params.glwe_dim.offset
    bad[off:off + 4] = np.frombuffer(np.int32(0x00FF0001).tobytes(), dtype=np.uint8)
    assert not lib.fhes_client_key_deserialize(ptr(bad), len(bad))
    assert "invalid parameters" in lib.fhes_last_error().decode()
    # a ciphertext of another width under this key: refused, not read past its end
    wide = ct_blob.copy()
    words = int(np.frombuffer(ct_blob[16:24].tobytes(), dtype=np.int64)[0])
    n_blocks = int(np.frombuffer(ct_blob[12:16].tobytes(), dtype=np.int32)[0])
    narrow = np.concatenate([ct_blob[:16], np.frombuffer(np.int64(words // 2).tobytes(), dtype=np.uint8),
                             ct_blob[24:24 + 2 * n_blocks + n_blocks * (words // 2) * 8]])
    c = lib.fhes_ct_deserialize(ptr(narrow), len(narrow))
    assert c, lib.fhes_last_error()
    vals = np.zeros(n_blocks, dtype=np.uint8)
    with pytest.raises(FheError):
        check(lib.fhes_ct_decrypt_blocks(ck, c, ptr(vals)))
    lib.fhes_ct_destroy(c)
    del wide
    # key generation refuses a length bound that makes the usize width meaningless
    assert not lib.fhes_client_key_generate(C.byref(p), 0, 1)
    assert not lib.fhes_client_key_generate(C.byref(p), 2 ** 40, 1)
    lib.fhes_ct_destroy(s)
    lib.fhes_client_key_destroy(ck)
