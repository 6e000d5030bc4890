"""Serialisation through the device.  The reference derives serde on its keys and
ciphertexts (`/root/reference/src/server_key.rs:11`, `client_key.rs:15`, `ciphertext/mod.rs:30,42,401`): a server receives a
ServerKey blob and FheString / FheUsize blobs, deserialises them and computes.  Here: the blobs of one key set go through
`fhes_server_key_deserialize` (uploads to the device, no client key) and `fhes_ct_deserialize` in a FRESH context, `find` /
`contains` / `splitn` run on the restored objects, and every word of every result ciphertext equals the run on the original,
never-serialised objects."""
import numpy as np
import pytest

from fhe_string_amd import api

pytestmark = pytest.mark.gpu

TEXT = "the serialised fox jumps over the lazy dog, twice: fox"
SEED = 0xF4E50000


def result_bytes(r):
    if isinstance(r, api.FheOption):
        return [r.is_some.to_bytes(), r.val.to_bytes()]
    if isinstance(r, api.FheStringSliceVector):
        return [p.to_bytes() for p in r._parts()]
    return [r.to_bytes()]


def run_ops(sk, a, pat, miss, n):
    return {"find": a.find(sk, pat), "find_miss": a.find(sk, miss), "contains": a.contains(sk, pat),
            "contains_miss": a.contains(sk, miss), "splitn": a.splitn(sk, n, pat)}


def test_server_key_and_ciphertext_blobs_run_on_the_device(libs):
    ck = api.ClientKey(api.default_params(), 255, SEED)
    a, pat, miss = api.FheString.new(ck, TEXT), api.FheString.new(ck, "fox"), api.FheString.new(ck, "cat", 5)
    n = api.FheUsize.new(ck, 2)
    blobs = {"ck": ck.to_bytes(), "sk": api.ServerKey.serialize(ck), "a": a.to_bytes(), "pat": pat.to_bytes(),
             "miss": miss.to_bytes(), "n": n.to_bytes()}
    assert blobs["sk"][:4] == b"FHES" and len(blobs["sk"]) > 100_000_000   # 170 MB: standard-domain BSK + KSK

    # the never-serialised run
    sk = api.ServerKey(ck)
    direct = run_ops(sk, a, pat, miss, n)
    direct_bytes = {k: result_bytes(v) for k, v in direct.items()}
    assert direct["find"].decrypt(ck) == TEXT.find("fox") and direct["find_miss"].decrypt(ck) is None
    assert ck.decrypt_bool(direct["contains"]) is True and ck.decrypt_bool(direct["contains_miss"]) is False
    assert direct["splitn"].decrypt(ck) == TEXT.split("fox", 1)
    sk.close()
    del direct, sk

    # a fresh context from the blobs alone; decryption with a client key restored from ITS blob
    sk2 = api.ServerKey.from_bytes(blobs["sk"])
    assert (sk2.num_blocks_char, sk2.num_blocks_usize) == (ck.num_blocks_char, ck.num_blocks_usize)
    assert bytes(sk2.params) == bytes(ck.params)
    a2, pat2, miss2 = (api.FheString.from_bytes(blobs[k]) for k in ("a", "pat", "miss"))
    n2 = api.FheUsize.from_bytes(blobs["n"])
    ck2 = api.ClientKey.from_bytes(blobs["ck"])
    assert a2.decrypt(ck2) == TEXT and n2.decrypt(ck2) == 2
    restored = run_ops(sk2, a2, pat2, miss2, n2)
    assert restored["find"].decrypt(ck2) == TEXT.find("fox") and restored["find_miss"].decrypt(ck2) is None
    assert ck2.decrypt_bool(restored["contains"]) is True and ck2.decrypt_bool(restored["contains_miss"]) is False
    assert restored["splitn"].decrypt(ck2) == TEXT.split("fox", 1)
    for op, parts in direct_bytes.items():
        got = result_bytes(restored[op])
        assert len(got) == len(parts)
        for i, (x, y) in enumerate(zip(got, parts)):
            assert x == y, f"{op}: part {i} of the result differs between the serialised and the direct run"

    # results are ciphertext objects like any other: a result blob restored and used as an operand
    idx = api.FheUsize.from_bytes(restored["find"].val.to_bytes())
    tail = a2.substr_from(sk2, idx)
    assert tail.decrypt(ck2) == TEXT[TEXT.find("fox"):]
    sk2.close()


def test_corrupted_server_key_blob_is_refused_on_a_machine_with_a_device(libs):
    from fhe_string_amd._lib import FheError
    ck = api.ClientKey(api.toy_params(), 63, 7)
    blob = bytearray(api.ServerKey.serialize(ck))
    with pytest.raises(FheError):
        api.ServerKey.from_bytes(bytes(blob[:-16]))
    bad = bytearray(blob)
    bad[8] ^= 0xFF                                   # kind
    with pytest.raises(FheError):
        api.ServerKey.from_bytes(bytes(bad))
    sk = api.ServerKey.from_bytes(bytes(blob))       # and the intact blob still loads
    s = api.FheString.new(ck, "abcabc")
    assert s.find(sk, api.FheString.new(ck, "ca")).decrypt(ck) == 2
    sk.close()


def test_a_server_that_received_its_key_can_serialise_it_again(libs):
    """`#[derive(Serialize, Deserialize)] struct ServerKey` (/root/reference/src/server_key.rs:11): the SERVER's object
    serialises, with no client key in reach.  deserialise -> serialise gives identical bytes, a key made
    from a client key on this machine gives the client-side blob's bytes, and the re-serialised key still computes."""
    ck = api.ClientKey(api.default_params(), 255, SEED)
    client_blob = api.ServerKey.serialize(ck)            # made where the key was generated: no device involved
    sk = api.ServerKey.from_bytes(client_blob)           # the server: all it ever sees is this blob
    again = sk.to_bytes()
    assert again == client_blob, "deserialise -> serialise changed the blob"
    direct = api.ServerKey(ck)                           # a key created from the client key serialises to the same bytes
    assert direct.to_bytes() == client_blob
    direct.close()
    sk.close()
    sk3 = api.ServerKey.from_bytes(again)
    a, pat = api.FheString.new(ck, TEXT), api.FheString.new(ck, "lazy")
    assert a.find(sk3, pat).decrypt(ck) == TEXT.find("lazy")
    sk3.close()


def test_clone_stays_on_the_device(libs):
    """`Clone` of a ciphertext object (compare.rs:155-160, mod.rs:250-263, split.rs:322-325 clone freely): identical words,
    independent rows, and not one byte over PCIe."""
    import ctypes as C
    from fhe_string_amd._lib import Stats, check, load
    lib = load()
    ck = api.ClientKey(api.default_params(), 255, SEED)
    sk = api.ServerKey(ck)
    a, pat = api.FheString.new(ck, TEXT), api.FheString.new(ck, "fox")
    host_only = a.clone()                                # before any operation the object lives in host memory: copied there
    assert host_only.to_bytes() == a.to_bytes()
    r = a.find(sk, pat)                                  # operands are on the device now, the result was born there
    ctx = sk.ctx

    def moved():
        st = Stats()
        check(lib.fhes_get_stats(ctx, C.byref(st)))
        return int(st.host_bytes_up), int(st.host_bytes_down)

    before = moved()
    a2, idx2 = a.clone(), r.val.clone()
    assert moved() == before, "a device-side clone moved bytes over PCIe"
    assert a2.to_bytes() == a.to_bytes() and idx2.to_bytes() == r.val.to_bytes()
    assert moved()[1] > before[1]                        # (the comparison itself downloads: the counter does count)
    # independent objects: the clone outlives its source and is an operand like any other
    want = r.val.decrypt(ck)
    del a, r
    assert a2.decrypt(ck) == TEXT and idx2.decrypt(ck) == want == TEXT.find("fox")
    assert a2.find(sk, pat).decrypt(ck) == want
    sk.close()
