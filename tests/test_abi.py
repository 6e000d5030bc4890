"""The C-ABI library loads and exports every symbol include/fhe_string_amd.h declares;
host-only entry points work and device entry points fail loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from fhe_string_amd import _lib
from fhe_string_amd._lib import Params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "fhe_string_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fhes_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(libs):
    lib = C.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) > 40
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # and the Python binding knows every one of them
    assert not [n for n in names if n not in _lib.SIGNATURES]


def test_parameter_sets_and_block_counts(libs):
    lib, _ = libs
    p = Params()
    assert lib.fhes_params_default(C.byref(p)) == 0
    assert (p.lwe_dim, p.glwe_dim, p.poly_size, p.pbs_base_log, p.pbs_level, p.ks_base_log, p.ks_level) == \
           (742, 1, 2048, 23, 1, 3, 5)
    assert lib.fhes_ct_words(C.byref(p)) == 2049
    assert lib.fhes_bsk_words(C.byref(p)) * 8 == 48627712
    assert lib.fhes_ksk_words(C.byref(p)) * 8 == 60866560
    for g, n in ((2, 764), (3, 765)):                           # multi-bit presets (config 5)
        m = Params()
        assert lib.fhes_params_multibit(C.byref(m), g) == 0
        assert (m.lwe_dim, m.grouping_factor, m.poly_size) == (n, g, 2048)
        assert lib.fhes_bsk_words(C.byref(m)) == (n // g) * (1 << g) * 4 * 2048
    assert lib.fhes_params_multibit(C.byref(Params()), 4) < 0
    bad = Params()
    lib.fhes_params_multibit(C.byref(bad), 2)
    bad.lwe_dim = 765                                           # not a multiple of the grouping factor
    assert not lib.fhes_client_key_generate(C.byref(bad), 255, 1)
    for l, bu in [(255, 4), (256, 5), (512, 5), (1024, 6), (3, 1), (63, 3)]:   # lib.rs:51-62
        ck = lib.fhes_client_key_generate(C.byref(p), l, 7)
        assert lib.fhes_client_key_num_blocks_char(ck) == 4
        assert lib.fhes_client_key_num_blocks_usize(ck) == bu
        lib.fhes_client_key_destroy(ck)


def test_string_validation_errors(libs):
    """the error cases of FheString::new (mod.rs:59-70)"""
    from fhe_string_amd.api import ClientKey, FheString, toy_params
    from fhe_string_amd._lib import FheError
    ck = ClientKey(toy_params(), l=15, seed=3)
    assert FheString.new(ck, "abc").num_blocks == 16
    assert FheString.new(ck, "abc", 7).num_blocks == 32
    assert FheString.new(ck, "abc", 7).decrypt(ck) == "abc"
    for s, pad, code in [("\xe9", None, -10), ("a" * 16, None, -12), ("a", 16, -13), ("abc", 2, -14)]:
        with pytest.raises(FheError) as e:
            FheString.new(ck, s, pad)
        assert e.value.code == code


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_device_entry_points_fail_loudly_without_gpu(libs):
    lib, _ = libs
    p = Params()
    lib.fhes_params_default(C.byref(p))
    assert not lib.fhes_ctx_create(C.byref(p), 0)
    assert b"no HIP device" in lib.fhes_last_error() or b"HIP error" in lib.fhes_last_error()
    ck = lib.fhes_client_key_generate(C.byref(p), 255, 7)
    assert not lib.fhes_server_key_create(ck, 0)
    lib.fhes_client_key_destroy(ck)


def test_compile_rejects_bad_requests(libs):
    lib, _ = libs
    p = Params()
    lib.fhes_params_default(C.byref(p))
    q = Params()
    lib.fhes_params_multibit(C.byref(q), 2)
    q.pbs_base_log = 18        # Level 0 runs it; its bootstrap noise leaves no room for the compiled circuits
    sh2 = np.array([3, 2], dtype=np.int32)
    assert not lib.fhes_program_compile(C.byref(q), b"find", sh2.ctypes.data_as(C.c_void_p), 2, 4)
    assert b"leaves no room to combine bootstrapped blocks" in lib.fhes_last_error() and b"base 2^18" in lib.fhes_last_error()
    sh = np.array([3, 2], dtype=np.int32)
    assert not lib.fhes_program_compile(C.byref(p), b"no_such_op", sh.ctypes.data_as(C.c_void_p), 2, 4)
    assert b"unknown operation" in lib.fhes_last_error()
    assert not lib.fhes_program_compile(C.byref(p), b"find", sh.ctypes.data_as(C.c_void_p), 1, 4)


def test_serialisation_round_trips(libs):
    from fhe_string_amd.api import ClientKey, FheString, FheUsize, toy_params
    ck = ClientKey(toy_params(), l=63, seed=9)
    s = FheString.new(ck, "hello", 7)
    blob = s.to_bytes()
    assert blob[:4] == b"FHES" and len(blob) == 24 + 2 * 32 + 32 * 257 * 8    # header | (degree, noise) per block | words
    assert blob[24:24 + 64] == bytes([3, 0] * 32)                               # fresh encryption: degree 3, noise level 0
    s2 = FheString.from_bytes(blob)
    lib = libs[0]
    deg, noi = (C.c_uint8 * 32)(), (C.c_uint8 * 32)()
    assert lib.fhes_ct_block_meta(s2._h, deg, noi) == 0 and list(deg) == [3] * 32 and list(noi) == [0] * 32
    ck2 = ClientKey.from_bytes(ck.to_bytes())
    assert s2.decrypt(ck2) == "hello" and s2.num_blocks == 32
    assert ck2.num_blocks_usize == ck.num_blocks_usize == 3
    assert FheUsize.from_bytes(FheUsize.new(ck, 37).to_bytes()).decrypt(ck2) == 37
    import pytest
    from fhe_string_amd._lib import FheError
    with pytest.raises(FheError):
        FheString.from_bytes(blob[:100])
    with pytest.raises(FheError):
        ClientKey.from_bytes(b"nope" + blob[4:])
    # production mode (seed 0): 256-bit OS keys; two keys differ, encryption randomness is not in the blob,
    # two encryptions of the same text differ, a restored key decrypts both
    k1, k2 = ClientKey(toy_params(), l=63), ClientKey(toy_params(), l=63)
    b1 = k1.to_bytes()
    assert b1 != k2.to_bytes() and len(b1) == len(ck.to_bytes()) + 16         # mode word + 32-byte key instead of seed + counter
    e1, e2 = FheString.new(k1, "same", 6), FheString.new(k1, "same", 6)
    assert e1.to_bytes() != e2.to_bytes()
    k1r = ClientKey.from_bytes(b1)
    assert e1.decrypt(k1r) == e2.decrypt(k1r) == "same"
    assert k1.to_bytes() == b1                                                # encrypting changed nothing that is serialised
    # ServerKey blob (server_key.rs:11): header + standard-domain BSK + KSK; deserialising needs a device
    p = toy_params()
    n = lib.fhes_client_key_serialize_server_key(ck._h, None, 0)
    assert n == 12 + C.sizeof(Params) + 8 + 4 + 4 + 8 + 8 + 8 * (lib.fhes_bsk_words(C.byref(p)) + lib.fhes_ksk_words(C.byref(p)))
    buf = (C.c_uint8 * n)()
    assert lib.fhes_client_key_serialize_server_key(ck._h, buf, n) == n and bytes(buf[:4]) == b"FHES"
    assert not lib.fhes_server_key_deserialize(buf, n, 0)                     # no HIP device here: fails loudly
    assert not lib.fhes_server_key_deserialize(buf, n - 8, 0) and b"truncated" in lib.fhes_last_error()


def test_scratch_cells_are_reused(libs):
    """node results share arena cells once all readers of the previous occupant ran in an earlier
    level (program.hip: flatten); results keep their own cells"""
    lib, _ = libs
    from tests.progs import Prog
    p = Params()
    lib.fhes_params_default(C.byref(p))
    for op, shape, bu in (("find", [65, 5], 4), ("split", [33, 2], 4), ("replace", [9, 3, 2, 8], 4)):
        pr = Prog(p, op, shape, bu)
        cells = lib.fhes_program_scratch_cells(pr.h)
        assert 0 < cells <= pr.n_nodes
        # the widest level is a lower bound, the node count an upper bound
        widest = max(int(pr.level_ptr[i + 1] - pr.level_ptr[i]) for i in range(pr.n_levels))
        assert cells >= widest
    assert lib.fhes_program_scratch_cells(Prog(p, "find", [65, 5], 4).h) < 1000   # 1665 nodes


def test_integration_doc_matches_the_header(libs):
    """INTEGRATION.md's `FhesParams` (the Rust binding a maintainer copies) is the header's fhes_params,
    field for field -- a stale doc here silently shifts every field after the gap -- and every extern
    function the doc declares exists in the header; the ctypes mirror has the same fields too."""
    header = open(os.path.join(ROOT, "include", "fhe_string_amd.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    body = re.search(r"typedef struct fhes_params \{(.*?)\} fhes_params;", header, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    c_fields = [(m.group(2), {"int32_t": "i32", "double": "f64"}[m.group(1)]) for m in re.finditer(r"(int32_t|double)\s+(\w+)\s*;", body)]
    rust = re.search(r"pub struct FhesParams \{(.*?)\}", doc, flags=re.S).group(1)
    r_fields = [(m.group(1), m.group(2)) for m in re.finditer(r"pub (\w+): (\w+)", rust)]
    assert c_fields == r_fields and len(c_fields) == 12
    assert [n for n, _ in c_fields] == [n for n, _ in Params._fields_]
    assert C.sizeof(Params) == 56 and "sizeof = 56" in header.replace("(sizeof = 56)", "sizeof = 56")
    declared = set(declared_symbols())
    for fn in re.findall(r"pub fn (fhes_\w+)\(", doc):
        assert fn in declared, f"INTEGRATION.md declares {fn}, the header does not"


def test_no_librccl_in_reach_is_a_clean_error_not_a_crash():
    """with no loadable librccl the error message was built from a second dlerror() call (NULL) -> segfault.
    The binding happens once per process, so the probe runs in a child with FHES_RCCL_LIB pointed at nothing."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from fhe_string_amd import _lib\n"
        "lib = _lib.load()\n"
        "buf = (C.c_uint8 * 128)()\n"
        "rc = lib.fhes_comm_unique_id(buf)\n"
        "assert rc < 0, rc\n"
        "msg = lib.fhes_last_error().decode()\n"
        "assert 'librccl' in msg and 'no-such-dir' in msg, msg\n"
        "assert lib.fhes_rccl_library_path() == b''\n"
        "print('clean:', msg)\n")
    env = dict(os.environ, FHES_RCCL_LIB="/no-such-dir/librccl.so.1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-1500:])
    assert "clean:" in r.stdout
