"""bindings/rust is source only (no cargo / rustc in the image), so it is checked mechanically instead of compiled
(VERDICT r3 item 8):
  * src/ffi.rs is exactly what gen_ffi.py derives from include/fhe_string_amd.h, and covers every exported symbol;
  * every `pub fn` of the reference crate (/root/reference/src/lib.rs:34-89, client_key.rs, ciphertext/*.rs; the list is the
    fixture tests/golden/reference_api.json, regenerated from /root/reference when that exists) has a same-named `pub fn` in the
    same file and impl context of the binding, with the same number of arguments;
  * every operation the binding asks the engine for exists in fhes_op_signature with matching operand counts;
  * the one hand-copied constant equals fhes_params_default; braces balance; every `mod` has its file."""
import ctypes as C
import importlib.util
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "bindings", "rust")
SRC = os.path.join(CRATE, "src")
GOLD = os.path.join(ROOT, "tests", "golden")


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


api_tool = load(os.path.join(GOLD, "make_reference_api.py"), "make_reference_api")
gen_ffi = load(os.path.join(CRATE, "gen_ffi.py"), "gen_ffi")


def rust_files():
    out = []
    for root in (SRC, os.path.join(CRATE, "examples")):
        for d, _, files in os.walk(root):
            out += [os.path.join(d, f) for f in files if f.endswith(".rs")]
    out.append(os.path.join(CRATE, "build.rs"))
    return sorted(out)


def test_ffi_rs_is_generated_from_the_header_and_complete():
    from tests.test_abi import declared_symbols
    text = open(os.path.join(SRC, "ffi.rs")).read()
    assert text == gen_ffi.generate(), "bindings/rust/src/ffi.rs is stale: run python3 bindings/rust/gen_ffi.py"
    declared = set(declared_symbols())
    in_ffi = set(re.findall(r"pub fn (fhes_\w+)\(", text))
    assert in_ffi == declared, (sorted(declared - in_ffi), sorted(in_ffi - declared))
    # spot checks of the type mapping
    assert "pub fn fhes_string_op(sk: *mut FhesServerKey, op: *const c_char, args: *const *mut FhesCt, n_args: i32, iargs: *const i64, " \
           "n_iargs: i32, outs: *mut *mut FhesCt, max_outs: i32, n_outs: *mut i32) -> c_int;" in text
    assert "pub fn fhes_client_key_generate(p: *const FhesParams, l: u64, seed: u64) -> *mut FhesClientKey;" in text
    assert "pub fn fhes_last_error() -> *const c_char;" in text
    assert "pub fn fhes_comm_unique_id(id: *mut u8) -> c_int;" in text and "id: *const u8) -> *mut FhesComm;" in text
    assert "pub lwe_noise_std: f64," in text and "pub const FHES_ERR_PAD_TOO_SHORT: i32 = -14;" in text


def test_reference_api_fixture_is_current():
    ref = "/root/reference/src"
    if not os.path.isdir(ref):
        pytest.skip("the reference is not on this machine: the committed fixture stands")
    want = api_tool.collect(ref)
    have = json.load(open(os.path.join(GOLD, "reference_api.json")))["functions"]
    assert want == have, "tests/golden/reference_api.json is stale: run python3 tests/golden/make_reference_api.py"


def test_every_public_function_of_the_reference_exists_in_the_binding():
    ref = json.load(open(os.path.join(GOLD, "reference_api.json")))["functions"]
    assert len(ref) == 85
    mine = {(e["file"], e["context"], e["name"]): e["arity"] for e in api_tool.collect(SRC, skip=())}
    missing, wrong = [], []
    for e in ref:
        key = (e["file"], e["context"], e["name"])
        if key not in mine:
            missing.append(key)
        elif mine[key] != e["arity"]:
            wrong.append((key, e["arity"], mine[key]))
    assert not missing, f"not in bindings/rust (same file, same impl): {missing}"
    assert not wrong, f"argument count differs (reference, binding): {wrong}"


def test_every_operation_the_binding_requests_exists_with_matching_operands(libs):
    lib = libs[0]
    ops_seen = set()
    for path in rust_files():
        src = open(path).read()
        for m in re.finditer(r'\bop1?\(\s*k,\s*(?:if reverse \{ "(\w+)" \} else \{ "(\w+)" \}|"(\w+)"|name)\s*,\s*&\[([^\]]*)\]\s*,\s*&\[([^\]]*)\]', src):
            names = [n for n in (m.group(1), m.group(2), m.group(3)) if n]
            n_ct = len([a for a in m.group(4).split(",") if a.strip()])
            n_int = len([a for a in m.group(5).split(",") if a.strip()])
            if not names:                     # split_family(k, name, p): the literals are checked below
                continue
            for name in names:
                sig = lib.fhes_op_signature(name.encode())
                assert sig, f"{os.path.basename(path)}: operation '{name}' is unknown to the engine"
                sig = sig.decode()
                want_ct = sum(3 if c == "V" else (0 if c == "I" else 1) for c in sig)
                assert (n_ct, n_int) == (want_ct, sig.count("I")), (path, name, sig, n_ct, n_int)
                ops_seen.add(name)
        for name in re.findall(r'self\.split_family\(k, "(\w+)", p\)', src):
            assert lib.fhes_op_signature(name.encode()) == b"SS", name
            ops_seen.add(name)
    # the whole operation table of the engine is reachable from the binding
    table = re.search(r"static const OpSig kOps\[\] = \{(.*?)\n\};", open(os.path.join(ROOT, "fhe_string_amd", "csrc", "circuit", "strings.cpp")).read(), flags=re.S).group(1)
    engine_ops = set(re.findall(r'\{"(\w+)", "\w+"\}', table))
    assert engine_ops - ops_seen == set(), f"operations no binding method reaches: {sorted(engine_ops - ops_seen)}"


def test_default_parameter_constant_equals_the_engines(libs):
    from fhe_string_amd._lib import Params
    p = Params()
    assert libs[0].fhes_params_default(C.byref(p)) == 0
    body = re.search(r"pub const PARAM_MESSAGE_2_CARRY_2_KS_PBS: ClassicPBSParameters = ClassicPBSParameters \{(.*?)\};",
                     open(os.path.join(SRC, "lib.rs")).read(), flags=re.S).group(1)
    fields = dict(re.findall(r"(\w+):\s*([-\d.e]+)", body))
    assert list(fields) == [n for n, _ in Params._fields_]
    for name, _ in Params._fields_:
        assert float(fields[name]) == float(getattr(p, name)), name


def test_gross_syntax():
    for path in rust_files():
        src = api_tool.strip(open(path).read())
        src = re.sub(r'"(?:[^"\\]|\\.)*"', '""', src)           # string literals
        src = re.sub(r"'(?:[^'\\]|\\.)'", "''", src)             # char literals
        for o, c in ("{}", "()", "[]"):
            assert src.count(o) == src.count(c), f"{path}: unbalanced {o}{c}"
        for m in re.finditer(r"^\s*(?:pub )?mod (\w+);", src, flags=re.M):
            d = os.path.dirname(path)
            base = os.path.splitext(os.path.basename(path))[0]
            cands = [os.path.join(d, m.group(1) + ".rs"), os.path.join(d, m.group(1), "mod.rs"),
                     os.path.join(d, base, m.group(1) + ".rs")]
            assert any(os.path.exists(c) for c in cands), f"{path}: mod {m.group(1)} has no file"
    toml = open(os.path.join(CRATE, "Cargo.toml")).read()
    assert 'name = "fhe_string"' in toml and 'links = "fhe_string_amd"' in toml
