"""The device's DECRYPTED OUTPUT BLOCKS against the oracle's own restatement of the string circuits.

`oracle/string_circuits.py` restates `find` (`/root/reference/src/ciphertext/search.rs:23-41`, `compare.rs:119-151`,
`logic.rs:41-65`, `mod.rs:315-355`), `contains` (`search.rs:17-20`), `eq` (`compare.rs:21-45`) and `split`
(`split.rs:238-341`) from the reference's files, over plain block values, without any code of `fhe_string_amd/csrc/circuit/`.
HIP-vs-oracle bit parity (tests/test_gpu_fullsize_oracle.py) runs the DAG the product's compiler emitted on both sides and cannot see
a wrong circuit; this test can: the compiled programs run on the device at the real parameter set on BASELINE configs 2-4 inputs, their
output ciphertexts are decrypted block by block and every block -- the index under a `None`, the slice bounds under `!is_some`
included -- must be the restatement's."""
import numpy as np
import pytest

from fhe_string_amd._lib import check
from oracle import string_circuits as sc
from tests.helpers import ptr, world
from tests.progs import Prog, decode_result, num_blocks_usize, string_blocks

pytestmark = pytest.mark.gpu


def device_blocks(w, op, texts, pads, bu):
    ins = [string_blocks(t, pd) for t, pd in zip(texts, pads)]
    p = Prog(w.p, op, [len(x) // 4 for x in ins], bu)
    cts = w.encrypt(np.concatenate(ins))
    ctx = w.ctx()
    h = check(w.lib.fhes_arena_alloc(ctx, len(cts)))
    check(w.lib.fhes_arena_upload(ctx, h, len(cts), ptr(cts)))
    in_h = np.arange(h, h + len(cts), dtype=np.int32)
    out_h = np.zeros(p.n_outputs, dtype=np.int32)
    check(w.lib.fhes_program_run(ctx, p.h, ptr(in_h), ptr(out_h)))
    dev = np.zeros((p.n_outputs, w.W), dtype=np.uint64)
    check(w.lib.fhes_arena_download(ctx, int(out_h[0]), p.n_outputs, ptr(dev)))
    check(w.lib.fhes_arena_free(ctx, h, len(cts)))
    check(w.lib.fhes_arena_free(ctx, int(out_h[0]), p.n_outputs))
    return [int(v) for v in w.decrypt(dev)], p


def config2_text():
    rng = np.random.default_rng(0xF4E50002)
    return "".join(chr(c) for c in rng.integers(0x20, 0x7F, size=64))


@pytest.mark.parametrize("form", ["", "_logdepth"])
def test_config2_find_blocks(form, libs):
    w = world("default", l=255)
    bu = num_blocks_usize(255)
    text = config2_text()
    for pat in (text[37:41], "\x7f\x7f\x7f\x7f", text[:4], text[60:64], ""):
        want, _ = sc.run_plain("find", [text, pat], [None, None], bu)
        got, p = device_blocks(w, "find" + form, [text, pat], [None, None], bu)
        assert got == want, (form, pat)
        exp = text.find(pat)
        assert decode_result("find", got, bu) == (None if exp < 0 else exp)


def test_config2_eq_and_contains_blocks(libs):
    w = world("default", l=255)
    bu = num_blocks_usize(255)
    text = config2_text()
    for op, b, pads in (("eq", text, [None, None]), ("eq", text[:63] + "~", [None, None]), ("eq", text[:40], [None, 64]),
                        ("contains", text[37:41], [None, None]), ("contains", "\x7f\x7f", [None, 6])):
        want, _ = sc.run_plain(op, [text, b], pads, bu)
        got, _ = device_blocks(w, op, [text, b], pads, bu)
        assert got == want, (op, b)


@pytest.mark.parametrize("form", ["", "_logdepth"])
def test_config4_contains_blocks(form, libs):
    w = world("default", l=1024)
    bu = num_blocks_usize(1024)
    rng = np.random.default_rng(0xF4E50004)
    s = "".join(chr(c) for c in rng.integers(0x20, 0x7F, size=1000))
    for pat, expect in ((s[700:708], 1), ("\x7f" * 8, 0)):
        want, _ = sc.run_plain("contains", [s, pat], [1024, None], bu)
        got, _ = device_blocks(w, "contains" + form, [s, pat], [1024, None], bu)
        assert got == want == [expect]


def test_config3_split_blocks(libs):
    """every one of the 258 x (1 + 2 x 5) output blocks of the log-depth program (the faithful program's blocks equal it on the
    plaintext model, tests/test_ref_circuits.py, and on the device word for word through the oracle, test_gpu_fullsize_oracle.py)"""
    w = world("default", l=256)
    bu = num_blocks_usize(256)
    rng = np.random.default_rng(0xF4E50003)
    chars = [chr(c) for c in rng.integers(0x61, 0x7B, size=256)]
    for i in range(7, 256, 8):
        chars[i] = ","
    s = "".join(chars)
    want, _ = sc.run_plain("split", [s, ","], [None, None], bu)
    got, p = device_blocks(w, "split_logdepth", [s, ","], [None, None], bu)
    assert len(got) == len(want) == 258 * 11
    bad = [i for i, (x, y) in enumerate(zip(got, want)) if x != y]
    assert not bad, f"{len(bad)} output blocks differ from the restatement, first {bad[:10]}"
    assert decode_result("split", got, bu, s) == s.split(",")


def test_faithful_split_blocks_on_a_shorter_input(libs):
    """the reference's serial circuit itself (split.rs:238-333 as written) on the device, block by block"""
    w = world("default", l=255)
    bu = num_blocks_usize(255)
    s = "ab,cde,,f,ghij,"
    for pat, pads in ((",", [None, None]), (",,", [20, 3]), ("", [None, None])):
        want, _ = sc.run_plain("split", [s, pat], pads, bu)
        got, _ = device_blocks(w, "split", [s, pat], pads, bu)
        assert got == want, pat


def device_blocks_args(w, op, args, bu):
    """like device_blocks, for operations with FheUsize / clear-integer operands (reference-table argument form)"""
    from fhe_string_amd import _lib
    from tests.progs import build_inputs
    shape, digits = build_inputs(_lib.load(), op, args, bu)
    p = Prog(w.p, op, shape, bu)
    cts = w.encrypt(np.asarray(digits, dtype=np.uint8))
    ctx = w.ctx()
    h = check(w.lib.fhes_arena_alloc(ctx, len(cts)))
    check(w.lib.fhes_arena_upload(ctx, h, len(cts), ptr(cts)))
    in_h = np.arange(h, h + len(cts), dtype=np.int32)
    out_h = np.zeros(p.n_outputs, dtype=np.int32)
    check(w.lib.fhes_program_run(ctx, p.h, ptr(in_h), ptr(out_h)))
    dev = np.zeros((p.n_outputs, w.W), dtype=np.uint64)
    check(w.lib.fhes_arena_download(ctx, int(out_h[0]), p.n_outputs, ptr(dev)))
    check(w.lib.fhes_arena_free(ctx, h, len(cts)))
    check(w.lib.fhes_arena_free(ctx, int(out_h[0]), p.n_outputs))
    return [int(v) for v in w.decrypt(dev)]


@pytest.mark.parametrize("form", ["", "_logdepth"])
def test_config5_suite_operations_blocks_on_a_shorter_input(form, libs):
    """the operations of BASELINE config 5's suite (find / rfind / split / replace / trim, `examples/cmd/main.rs:127-142`, `:219-238`,
    `:269-283`, `:442-449`) and their neighbours on a 24-character padded input: the device's decrypted output blocks, in both forms of
    every circuit, against the restatement (the check at device level that round 3's `replace_logdepth` fault would have failed)"""
    w = world("default", l=255)
    bu = num_blocks_usize(255)
    text = "  the fish and the fishes "
    A = {"s": text, "pad": 28}
    P, S = {"s": "fish", "pad": None}, {"s": "_", "pad": None}
    cases = [("rfind", [A, P]), ("trim", [A]), ("trim_start", [A]), ("trim_end", [A]), ("replace", [A, P, S, {"i": 30}]),
             ("replacen", [A, P, S, {"u": 1}, {"i": 30}]), ("splitn", [A, {"u": 2}, P]), ("rsplit", [A, P]),
             ("split_once", [A, P]), ("strip_suffix", [A, {"s": "es ", "pad": None}]), ("split_ascii_whitespace", [A])]
    for op, args in cases:
        want, _ = sc.run_plain_args(op, args, bu)
        try:
            got = device_blocks_args(w, op + form, args, bu)
        except Exception as e:      # an operation without a log-depth form
            assert form == "_logdepth" and "unknown" in str(e).lower(), (op, e)
            continue
        assert got == want, (op + form, [i for i, (x, y) in enumerate(zip(got, want)) if x != y][:10])
