"""The oracle's OWN restatement of the string circuits (oracle/string_circuits.py: written from the reference's source
files, no code shared with fhe_string_amd/csrc/circuit/) pinned against what pins the reference itself -- the tables of
/root/reference/src/ciphertext/tests/*.rs (tests/golden/reference_tables.json) --, on plain block values and under the
oracle's own encryption, and compared BLOCK BY BLOCK with the product's compiled programs on the plaintext model."""
import json
import os

import numpy as np
import pytest

from oracle import string_circuits as sc
from tests.progs import Prog, decode_result, num_blocks_usize, string_blocks

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "reference_tables.json")))["cases"]
COVERED = {"find", "rfind", "contains", "eq", "ne", "lt", "le", "gt", "ge", "is_empty", "len", "starts_with", "ends_with",
           "split", "rsplit", "split_inclusive", "split_terminator", "rsplit_terminator"}


def operands(case):
    texts, pads = [case["a"]], [case.get("a_pad")]
    if "b" in case and case["b"] is not None:
        texts.append(case["b"])
        pads.append(case.get("b_pad"))
    return texts, pads


def decode(case, blocks, bu):
    op = case["op"]
    if op in ("find", "rfind"):
        return decode_result("find", blocks, bu)
    if op == "len":
        return decode_result("len", blocks, bu)
    if "split" in op:
        return decode_result("split", blocks, bu, case["a"])
    return bool(blocks[0] & 1)


def test_every_table_operation_is_restated():
    assert {c["op"] for c in CASES} <= COVERED


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_reference_tables_on_plain_blocks(idx):
    case = CASES[idx]
    bu = num_blocks_usize(255)
    texts, pads = operands(case)
    blocks, n_lut = sc.run_plain(case["op"], texts, pads, bu)
    assert all(0 <= v < 4 for v in blocks), "results must be clean digits"
    assert decode(case, blocks, bu) == case["expect"], (case, blocks)


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_compiled_programs_produce_the_restatements_blocks(idx, libs):
    """the product's circuit compiler (faithful AND log-depth forms, plaintext model of the compiled program) against the
    restatement: every output block, including the ones a decoded result ignores (index under None, slices under !is_some)"""
    from fhe_string_amd._lib import FheError
    from tests.test_circuits_plain import prog
    case = CASES[idx]
    bu = num_blocks_usize(255)
    texts, pads = operands(case)
    want, _ = sc.run_plain(case["op"], texts, pads, bu)
    ins = [string_blocks(t, pd) for t, pd in zip(texts, pads)]
    for form in ("", "_logdepth"):
        try:
            p = prog(case["op"] + form, [len(x) // 4 for x in ins], bu)
        except (FheError, AssertionError):
            assert form == "_logdepth"      # an operation without a scan has no log-depth form
            continue
        got = [int(v) for v in p.run_plain(np.concatenate(ins).astype("int64"))]
        assert got == want, (case, form)


# ---- under the oracle's own encryption -------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ow(libs):
    from tests.helpers import OracleWorld
    return OracleWorld("toy")


def _cost(case):
    texts, pads = operands(case)
    return sc.run_plain(case["op"], texts, pads, num_blocks_usize(63))[1]


ENC_CASES = [i for i, c in enumerate(CASES) if c["op"] in ("find", "contains", "eq", "split", "rfind", "len", "starts_with", "lt")]


@pytest.mark.parametrize("idx", ENC_CASES)
def test_reference_tables_under_oracle_encryption(idx, ow):
    """keys, encryption and decoding from the oracle's client, every lookup an oracle_pbs (toy parameters, l = 63): the
    decrypted OUTPUT BLOCKS equal the plain-block run's, and the decoded value is the table's"""
    case = CASES[idx]
    bu = num_blocks_usize(63)
    texts, pads = operands(case)
    want, n_plain = sc.run_plain(case["op"], texts, pads, bu)
    if n_plain > 12000:
        pytest.skip(f"{n_plain} bootstraps: covered on plain blocks")
    got, n_enc = sc.run_encrypted(ow, case["op"], texts, pads, bu)
    assert n_enc == n_plain
    assert got == want
    assert decode(case, got, bu) == case["expect"]


def test_config1_at_the_real_parameter_set_under_oracle_encryption(libs):
    """BASELINE config 1 (README.md:34 of the reference): input " A bcbc ", pattern "bc": eq false, contains true, find Some(3),
    at PARAM_MESSAGE_2_CARRY_2_KS_PBS with real noise, every lookup a full-size oracle bootstrap"""
    from tests.helpers import OracleWorld
    real = OracleWorld("default", seed=0xF4E5C1)
    bu = num_blocks_usize(255)
    for op, expect in (("eq", False), ("contains", True), ("find", 3)):
        got, n = sc.run_encrypted(real, op, [" A bcbc ", "bc"], [None, None], bu)
        want, _ = sc.run_plain(op, [" A bcbc ", "bc"], [None, None], bu)
        assert got == want, op
        assert decode({"op": op, "a": " A bcbc "}, got, bu) == expect


# ---- the rest of the reference's tables (trim / replace / insert / convert / remaining split variants) ---------------------
CASES_EXT = json.load(open(os.path.join(HERE, "golden", "reference_tables.json")))["cases_ext"]


@pytest.mark.parametrize("idx", range(len(CASES_EXT)))
def test_extended_tables_on_plain_blocks_and_against_the_compiled_programs(idx, libs):
    """every table case of tests/{trim,replace,insert,convert,split,compare}.rs beyond the hot path: the restatement's decoded
    result is the table's, and the compiled programs (faithful AND log-depth) produce the restatement's OUTPUT BLOCKS one by one"""
    from fhe_string_amd import _lib
    from fhe_string_amd._lib import FheError
    from tests.progs import build_inputs, decode_generic
    from tests.test_circuits_plain import prog
    case = CASES_EXT[idx]
    bu = num_blocks_usize(255)
    want, _ = sc.run_plain_args(case["op"], case["args"], bu)
    shape, digits = build_inputs(_lib.load(), case["op"], case["args"], bu)
    p = prog(case["op"], shape, bu)
    self_str = case["args"][0]["s"] if "s" in case["args"][0] else None
    assert decode_generic(case["op"], np.array(want), p.out_parts, bu, self_str) == case["expect"], (case, want)
    for form in ("", "_logdepth"):
        try:
            pf = prog(case["op"] + form, shape, bu)
        except (FheError, AssertionError):
            assert form == "_logdepth"
            continue
        got = [int(v) for v in pf.run_plain(digits.astype("int64"))]
        assert got == want, (case, form)


def test_random_inputs_block_by_block_against_both_forms_of_the_compiled_programs(libs):
    """random strings, occurring patterns of 1-7 characters (the class of the round-3 fault: `replace_logdepth` lost patterns of 4-5
    characters), padded and unpadded: restatement == faithful program == log-depth program, every output block"""
    import random
    from fhe_string_amd import _lib
    from fhe_string_amd._lib import FheError
    from tests.progs import build_inputs
    from tests.test_circuits_plain import prog
    rng = random.Random(0x5EED4)
    bu = num_blocks_usize(255)
    alphabet = "ab c,x"
    checked = 0
    for trial in range(36):
        n = rng.randint(4, 14)
        text = "".join(rng.choice(alphabet) for _ in range(n))
        m = rng.randint(1, min(7, n))
        at = rng.randint(0, n - m)
        pat = text[at:at + m] if rng.random() < 0.8 else "".join(rng.choice(alphabet) for _ in range(m))
        sub = "".join(rng.choice("yz") for _ in range(rng.randint(0, 3)))
        pad_a = None if rng.random() < 0.5 else n + rng.randint(0, 3)
        pad_p = None if rng.random() < 0.6 else m + rng.randint(0, 2)
        A, P, S = {"s": text, "pad": pad_a}, {"s": pat, "pad": pad_p}, {"s": sub, "pad": None}
        ops = [("find", [A, P]), ("rfind", [A, P]), ("contains", [A, P]), ("split", [A, P]), ("rsplit", [A, P]),
               ("replace", [A, P, S, {"i": n + 6}]), ("replacen", [A, P, S, {"u": rng.randint(0, 3)}, {"i": n + 6}]),
               ("strip_suffix", [A, P]), ("split_once", [A, P]), ("trim", [A]), ("splitn", [A, {"u": rng.randint(0, 4)}, P])]
        for op, args in rng.sample(ops, 4):
            want, _ = sc.run_plain_args(op, args, bu)
            shape, digits = build_inputs(_lib.load(), op, args, bu)
            for form in ("", "_logdepth"):
                try:
                    pf = prog(op + form, shape, bu)
                except (FheError, AssertionError):
                    assert form == "_logdepth"
                    continue
                got = [int(v) for v in pf.run_plain(digits.astype("int64"))]
                assert got == want, (op + form, args)
                checked += 1
    assert checked >= 200
