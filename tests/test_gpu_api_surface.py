"""The entry points added in round 4 so that the Rust binding covers the reference's whole public surface: `ClientKey::encrypt` /
`decrypt` (`/root/reference/src/client_key.rs:23-33`), `FheAsciiChar::new` (`ciphertext/mod.rs:35`), `FheUsize::new_trivial` (`:426`),
and the crate-public helpers `any` / `all` / `if_then_else_*` (`ciphertext/logic.rs:13-65`) and `element_at_bool` (`mod.rs:271-287`) as
operations of their own -- on the device at the real parameter set."""
import pytest

from fhe_string_amd import api

pytestmark = pytest.mark.gpu


def test_new_entry_points_on_the_device(libs):
    ck, sk = api.generate_keys_with_params(api.default_params(), 255, seed=0xF4E50000)
    try:
        # ClientKey::encrypt / decrypt and FheAsciiChar::new
        r = ck.encrypt(201)
        assert r.num_blocks == ck.num_blocks_char and ck.decrypt(r) == 201
        c = api.FheAsciiChar.new(ck, ord("q"))
        assert c.decrypt(ck) == ord("q")
        assert ck.decrypt_bool(c.is_lowercase(sk)) is True and c.to_uppercase(sk).decrypt(ck) == ord("Q")
        # FheUsize::new_trivial as an operand
        s = api.FheString.new(ck, "trivial index", 20)
        assert s.char_at(sk, api.FheUsize.new_trivial(sk, 8)).decrypt(ck) == ord("i")
        assert api.FheUsize.new_trivial(sk, 200).decrypt(ck) == 200
        # logic.rs helpers
        flags = [ck_bool(ck, sk, s, ch) for ch in "txz"]          # "trivial index" has t and x, no z: BooleanBlocks computed on the device
        assert [ck.decrypt_bool(f) for f in flags] == [True, True, False]
        assert ck.decrypt_bool(api.any_of(sk, flags)) is True and ck.decrypt_bool(api.all_of(sk, flags)) is False
        assert ck.decrypt_bool(api.all_of(sk, flags[:2])) is True
        for i, want in ((0, True), (2, False), (3, False), (77, False)):
            assert ck.decrypt_bool(api.element_at_bool(sk, flags, api.FheUsize.new(ck, i))) is want
        n = api.FheUsize.new(ck, 123)
        assert api.if_then_else_zero(sk, flags[0], n).decrypt(ck) == 123 and api.if_then_else_zero(sk, flags[2], n).decrypt(ck) == 0
        assert api.scalar_if_then_else_zero(sk, flags[1], 45).decrypt(ck) == 45
        assert ck.decrypt_bool(api.if_then_else_bool(sk, flags[2], flags[0], flags[2])) is False
        assert ck.decrypt_bool(api.if_then_else_bool(sk, flags[0], flags[0], flags[2])) is True
    finally:
        sk.close()


def ck_bool(ck, sk, s, ch):
    """`s` contains `ch`, as an encrypted boolean"""
    return s.contains(sk, api.FheString.new(ck, ch))
