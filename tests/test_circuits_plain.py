"""Host logic: every compiled circuit evaluated on a plaintext model (values modulo 32 with
the negacyclic lookup rule) against the reference's own test tables and against Python
restatements of `std::str` on random ASCII inputs.  No crypto, no GPU."""
import json
import os
import random

import pytest

from fhe_string_amd.api import default_params
from tests.progs import Prog, decode_result, num_blocks_usize, string_blocks

HERE = os.path.dirname(os.path.abspath(__file__))
_TABLES = json.load(open(os.path.join(HERE, "golden", "reference_tables.json")))
CASES = _TABLES["cases"]
CASES_EXT = _TABLES["cases_ext"]

_cache = {}


def prog(op, shape, bu):
    key = (op, tuple(shape), bu)
    if key not in _cache:
        _cache[key] = Prog(default_params(), op, shape, bu)
    return _cache[key]


def run_plain(op, a, a_pad=None, b=None, b_pad=None, l=255):
    bu = num_blocks_usize(l)
    ins = [string_blocks(a, a_pad)]
    if b is not None:
        ins.append(string_blocks(b, b_pad))
    shape = [len(x) // 4 for x in ins]
    p = prog(op, shape, bu)
    import numpy as np
    outs = p.run_plain(np.concatenate(ins).astype("int64"))
    return decode_result(op, outs, bu, a), p


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c['op']}-{c['a']!r}-{c['b']!r}-{c['a_pad']}")
def test_reference_tables(case, libs):
    got, _ = run_plain(case["op"], case["a"], case["a_pad"], case["b"], case["b_pad"])
    assert got == case["expect"], case


def rust_find(a, b):
    i = a.find(b)
    return None if i < 0 else i


def rust_rfind(a, b):
    i = a.rfind(b)
    return None if i < 0 else i


def rust_split(a, p):
    return [""] + list(a) + [""] if p == "" else a.split(p)


def ref_ends_with(a, a_slots, b, b_slots):
    """bug-compatible model of search.rs:222-244 (terminator tested on `a`)"""
    A = list(a.encode()) + [0] * (a_slots - len(a))
    B = list(b.encode()) + [0] * (b_slots - len(b))
    for i in range(len(A) - 1):
        if all(x == y or x == 0 for x, y in zip(A[i:], B)):
            return True
    return False


def test_random_strings_against_std(libs):
    rnd = random.Random(0xF4E5)
    alphabet = "abxy ,"
    for trial in range(60):
        la, lb = rnd.randint(0, 9), rnd.randint(0, 3)
        a = "".join(rnd.choice(alphabet) for _ in range(la))
        b = "".join(rnd.choice(alphabet) for _ in range(lb))
        if trial % 3 == 0 and la >= lb and lb > 0:
            k = rnd.randint(0, la - lb)
            b = a[k:k + lb]
        a_pad = rnd.choice([None, la + rnd.randint(0, 3)])
        b_pad = rnd.choice([None, lb + rnd.randint(0, 2)])
        ctx = (a, a_pad, b, b_pad)
        # reference quirk (search.rs:34): a string with only its terminator slot has no start
        # position, so find/contains report "not found" even for the empty pattern
        no_positions = (la if a_pad is None else a_pad) == 0
        assert run_plain("find", a, a_pad, b, b_pad)[0] == (None if no_positions else rust_find(a, b)), ctx
        assert run_plain("rfind", a, a_pad, b, b_pad)[0] == (
            (0 if b == "" else None) if no_positions else rust_rfind(a, b)), ctx
        assert run_plain("contains", a, a_pad, b, b_pad)[0] == (False if no_positions else (b in a)), ctx
        assert run_plain("starts_with", a, a_pad, b, b_pad)[0] == a.startswith(b), ctx
        assert run_plain("eq", a, a_pad, b, b_pad)[0] == (a == b), ctx
        assert run_plain("ne", a, a_pad, b, b_pad)[0] == (a != b), ctx
        assert run_plain("lt", a, a_pad, b, b_pad)[0] == (a < b), ctx
        assert run_plain("le", a, a_pad, b, b_pad)[0] == (a <= b), ctx
        assert run_plain("gt", a, a_pad, b, b_pad)[0] == (a > b), ctx
        assert run_plain("ge", a, a_pad, b, b_pad)[0] == (a >= b), ctx
        assert run_plain("len", a, a_pad)[0] == len(a), ctx
        assert run_plain("is_empty", a, a_pad)[0] == (a == ""), ctx
        a_slots = (la if a_pad is None else a_pad) + 1
        b_slots = (lb if b_pad is None else b_pad) + 1
        assert run_plain("ends_with", a, a_pad, b, b_pad)[0] == ref_ends_with(a, a_slots, b, b_slots), ctx
        if b != "":  # the reference documents empty-pattern split as undefined (split.rs:337-338)
            assert run_plain("split", a, a_pad, b, b_pad)[0] == rust_split(a, b), ctx


def test_empty_and_ragged_inputs(libs):
    # edge cases the reference tests or notes
    assert run_plain("find", "", None, "", None)[0] is None      # find_all scans 0 positions (search.rs:34)
    assert run_plain("contains", "", 3, "", 2)[0] is True
    assert run_plain("find", "abc", None, "abcd", None)[0] is None
    assert run_plain("find", "abc", None, "abc", 7)[0] == 0      # pattern padded beyond the input
    assert run_plain("eq", "", None, "", 5)[0] is True
    assert run_plain("lt", "", None, "a", None)[0] is True
    assert run_plain("len", "", 6)[0] == 0


def test_north_star_shape(libs):
    """BASELINE config 2: 64-char unpadded input, 4-char pattern, l = 255"""
    import numpy as np
    rng = np.random.default_rng(0xF4E50002)
    s = "".join(chr(c) for c in rng.integers(0x20, 0x7F, size=64))
    pat = s[37:41]
    got, p = run_plain("find", s, None, pat, None)
    assert got == s.find(pat)
    assert run_plain("find", s, None, "\x7f\x7f\x7f\x7f", None)[0] is None
    assert p.n_inputs == (65 + 5) * 4 and p.n_outputs == 5
    assert run_plain("contains", s, None, pat, None)[0] is True


# ---- operations with FheUsize / clear-int operands or string results -----------------------
def run_ext(op, args, l=255):
    import numpy as np
    from fhe_string_amd import _lib
    from tests.progs import build_inputs, decode_generic
    bu = num_blocks_usize(l)
    shape, digits = build_inputs(_lib.load(), op, args, bu)
    p = prog(op, shape, bu)
    outs = p.run_plain(digits.astype("int64"))
    self_str = args[0]["s"] if "s" in args[0] else None
    return decode_generic(op, outs, p.out_parts, bu, self_str), p


@pytest.mark.parametrize("case", CASES_EXT, ids=lambda c: f"{c['op']}-{c['args'][0].get('s')!r}-{len(c['args'])}")
def test_reference_tables_ext(case, libs):
    got, _ = run_ext(case["op"], case["args"])
    assert got == case["expect"], case


def S(s, pad=None):
    return {"s": s, "pad": pad}


def test_index_operations_against_python(libs):
    rnd = random.Random(7)
    for trial in range(25):
        la = rnd.randint(0, 7)
        a = "".join(rnd.choice("abc xy") for _ in range(la))
        pad = rnd.choice([None, la + rnd.randint(0, 2)])
        i, j = rnd.randint(0, la + 2), rnd.randint(0, la + 2)
        ctx = (a, pad, i, j)
        assert run_ext("char_at", [S(a, pad), {"u": i}])[0] == (ord(a[i]) if i < la else 0), ctx
        assert run_ext("substr_to", [S(a, pad), {"u": i}])[0] == a[:i], ctx
        assert run_ext("truncate", [S(a, pad), {"u": i}])[0] == a[:i], ctx
        assert run_ext("substr_from", [S(a, pad), {"u": i}])[0] == a[i:], ctx
        assert run_ext("substr_end", [S(a, pad), {"u": i}, {"u": j}])[0] == a[i:j], ctx
        b = "".join(rnd.choice("ab") for _ in range(rnd.randint(0, 3)))
        k = rnd.randint(0, la)
        assert run_ext("insert", [S(a, pad), {"u": k}, S(b)])[0] == a[:k] + b + a[k:], ctx + (b, k)
        assert run_ext("add", [S(a, pad), S(b)])[0] == a + b, ctx + (b,)


def test_replace_and_split_variants_against_python(libs):
    rnd = random.Random(11)
    for trial in range(20):
        la = rnd.randint(1, 7)
        a = "".join(rnd.choice("abx") for _ in range(la))
        p = "".join(rnd.choice("abx") for _ in range(rnd.randint(1, 2)))
        # the reference's replace emits one output char per step and therefore mishandles an EMPTY
        # replacement (it spends a slot on the terminator of `s`, replace.rs:91-94): not tested there
        r = "".join(rnd.choice("yz") for _ in range(rnd.randint(1, 2)))
        n = rnd.randint(0, 3)
        res = a.replace(p, r)
        ctx = (a, p, r, n)
        assert run_ext("replace", [S(a), S(p), S(r), {"i": len(res)}])[0] == res, ctx
        resn = a.replace(p, r, n)
        assert run_ext("replacen", [S(a), S(p), S(r), {"u": n}, {"i": len(resn)}])[0] == resn, ctx
        assert run_ext("splitn", [S(a), {"u": n}, S(p)])[0] == (a.split(p, n - 1) if n else []), ctx
        i = a.find(p)
        assert run_ext("split_once", [S(a), S(p)])[0] == ([a[:i], a[i + len(p):]] if i >= 0 else None), ctx
        assert run_ext("strip_prefix", [S(a), S(p)])[0] == (a[len(p):] if a.startswith(p) else None), ctx
        w = "".join(rnd.choice("a b\t") for _ in range(rnd.randint(0, 7)))
        assert run_ext("split_ascii_whitespace", [S(w)])[0] == w.split(), (w,)
        assert run_ext("trim", [S(w)])[0] == w.strip(" \t"), (w,)
        m = "".join(rnd.choice("aBcD zZ@[`{") for _ in range(rnd.randint(0, 6)))
        assert run_ext("to_lowercase", [S(m)])[0] == m.lower(), (m,)
        assert run_ext("to_uppercase", [S(m)])[0] == m.upper(), (m,)


def test_logdepth_find_equals_faithful_find(libs):
    rnd = random.Random(5)
    for trial in range(40):
        la, lb = rnd.randint(0, 12), rnd.randint(0, 3)
        a = "".join(rnd.choice("ab") for _ in range(la))
        b = "".join(rnd.choice("ab") for _ in range(lb))
        a_pad = rnd.choice([None, la + rnd.randint(0, 3)])
        assert run_plain("find_logdepth", a, a_pad, b, None)[0] == run_plain("find", a, a_pad, b, None)[0], (a, b)
    _, p_fast = run_plain("find_logdepth", "x" * 64, None, "xxxx", None)
    _, p_ref = run_plain("find", "x" * 64, None, "xxxx", None)
    assert p_fast.n_levels < 20 < p_ref.n_levels


def test_split_family_matches_the_references_own_algorithm(libs):
    """bug-compatible parity: random inputs, including overlapping patterns and matches at
    position 0 where the reference's rsplit differs from std (search.rs:114-152)"""
    from tests.ref_model import ref_split
    rnd = random.Random(23)
    n_std_diff = 0
    for trial in range(60):
        la = rnd.randint(0, 7)
        a = "".join(rnd.choice("xa") for _ in range(la))
        p = "".join(rnd.choice("xa") for _ in range(rnd.randint(1, 3)))
        a_pad = rnd.choice([None, la + rnd.randint(0, 2)])
        p_pad = rnd.choice([None, len(p) + rnd.randint(0, 1)])
        for op in ("split", "rsplit", "split_inclusive", "split_terminator", "rsplit_terminator"):
            got = run_plain(op, a, a_pad, p, p_pad)[0]
            assert got == ref_split(op, a, a_pad, p, p_pad), (op, a, a_pad, p, p_pad)
        n_std_diff += run_plain("rsplit", a, a_pad, p, p_pad)[0] != list(reversed(a.rsplit(p)))
    assert run_plain("rsplit", "xxa", None, "xx", None)[0] == ["xxa"]      # the reference's quirk, reproduced
    assert n_std_diff > 0


def test_slice_vector_methods_and_repeat(libs):
    """FheStringSliceVector::{len,get,truncate,reverse} (split.rs:30-113, 229) and repeat
    (insert.rs:20-44) against plain restatements of the reference's code"""
    from tests import ref_model as rm
    rnd = random.Random(5)
    n_some = 0
    for trial in range(12):
        a = "".join(rnd.choice("xab") for _ in range(rnd.randint(0, 6)))
        p = rnd.choice(["x", "xa", "a", ""])
        a_pad = rnd.choice([None, len(a) + 1])
        op = rnd.choice(["split", "rsplit", "split_inclusive"])
        v = rm.raw_split(op, a, a_pad, p, None)
        V = {"v": v}
        assert run_ext("slices_len", [V])[0] == rm.slices_len(v)
        i = rnd.randint(0, 3)
        assert run_ext("slices_truncate", [V, {"u": i}])[0] == [(int(x[0]), x[1], x[2]) for x in rm.slices_truncate(v, i)]
        assert run_ext("slices_truncate_logdepth", [V, {"u": i}])[0] == [(int(x[0]), x[1], x[2]) for x in rm.slices_truncate(v, i)]
        assert run_ext("slices_reverse", [V])[0] == [(int(x[0]), x[1], x[2]) for x in reversed(v)]
        for i in range(0, rm.slices_len(v) + 1):
            got = run_ext("slices_get", [V, S(a, a_pad), {"u": i}])[0]
            assert got == rm.slices_get(v, a, a_pad, i), (op, a, a_pad, p, i)
            assert run_ext("slices_get_logdepth", [V, S(a, a_pad), {"u": i}])[0] == got, (op, a, a_pad, p, i)   # prefix-count form
            n_some += got is not None
    assert n_some > 0
    for trial in range(10):
        a = "".join(rnd.choice("ab") for _ in range(rnd.randint(0, 4)))
        pad = rnd.choice([None, len(a) + 2])
        n, l = rnd.randint(0, 4), rnd.randint(0, 12)
        want = (a * n)[:l]
        assert run_ext("repeat", [S(a, pad), {"u": n}, {"i": l}])[0] == want, (a, pad, n, l)


def test_remaining_public_methods(libs):
    """substr_eq (compare.rs:119), FheAsciiChar predicates / case (convert.rs:14-49, trim.rs:15-25),
    FheUsize::new_from_bool (mod.rs:431)"""
    rnd = random.Random(11)
    for a, i, b in [("abcabc", 0, "abc"), ("abcabc", 3, "abc"), ("abcabc", 1, "abc"), ("abc", 2, "c"), ("abc", 3, ""),
                    ("abc", 1, "bcd"), ("", 0, ""), ("ab", 5, "a")]:
        want = a[i:i + len(b)] == b if i <= len(a) else (b == "")
        assert run_ext("substr_eq", [S(a), {"i": i}, S(b)])[0] == want, (a, i, b)
    assert run_ext("substr_eq", [S("abc", 6), {"i": 1}, S("bc", 4)])[0] is True
    for ch in [0, 9, 13, 14, 32, 64, 65, 90, 91, 96, 97, 122, 123, 127] + [rnd.randrange(128) for _ in range(10)]:
        c = chr(ch)
        assert run_ext("is_uppercase", [{"c": ch}])[0] == ("A" <= c <= "Z"), ch
        assert run_ext("is_lowercase", [{"c": ch}])[0] == ("a" <= c <= "z"), ch
        assert run_ext("is_whitespace", [{"c": ch}])[0] == (9 <= ch <= 13 or ch == 32), ch
        assert run_ext("char_to_lowercase", [{"c": ch}])[0] == ord(c.lower()), ch
        assert run_ext("char_to_uppercase", [{"c": ch}])[0] == ord(c.upper()), ch
    assert run_ext("usize_from_bool", [{"b": 1}])[0] == 1 and run_ext("usize_from_bool", [{"b": 0}])[0] == 0


# ---- opt-in log-depth circuits ("<op>_logdepth"): same results as the faithful circuits --------
LOGDEPTH_OPS = {"find", "rfind", "lt", "le", "gt", "ge", "split", "rsplit", "split_inclusive", "split_terminator",
                "rsplit_terminator", "splitn", "rsplitn", "split_once", "rsplit_once", "trim", "trim_start", "trim_end",
                "strip_prefix", "strip_suffix", "split_ascii_whitespace", "substr_from", "substr_end", "replace", "replacen"}


@pytest.mark.parametrize("case", [c for c in CASES if c["op"] in LOGDEPTH_OPS],
                         ids=lambda c: f"{c['op']}-{c['a']!r}-{c['b']!r}-{c['a_pad']}")
def test_logdepth_reference_tables(case, libs):
    got, _ = run_plain(case["op"] + "_logdepth", case["a"], case["a_pad"], case["b"], case["b_pad"])
    assert got == case["expect"], case


@pytest.mark.parametrize("case", [c for c in CASES_EXT if c["op"] in LOGDEPTH_OPS],
                         ids=lambda c: f"{c['op']}-{c['args'][0].get('s')!r}-{len(c['args'])}")
def test_logdepth_reference_tables_ext(case, libs):
    got, _ = run_ext(case["op"] + "_logdepth", case["args"])
    assert got == case["expect"], case


def test_logdepth_equals_faithful_on_random_inputs(libs):
    """every block of every output, not only the decoded value: the log-depth circuits must compute the
    same booleans and integers as the faithful ones (overlapping patterns, padded patterns up to the
    automaton's window of 7, empty pattern, matches at the ends)"""
    import numpy as np
    rnd = random.Random(0x10D)
    for trial in range(90):
        la = rnd.randint(0, 9) if trial < 70 else rnd.randint(7, 12)
        a = "".join(rnd.choice("xa") for _ in range(la))
        p = "".join(rnd.choice("xa") for _ in range(rnd.randint(0, 3)))
        if trial >= 70:    # occurring patterns of 4 .. 7 characters (the automaton's window goes to 7)
            lp = rnd.randint(4, 7)
            i0 = rnd.randint(0, la - lp)
            p = a[i0:i0 + lp]
        a_pad = rnd.choice([None, la + rnd.randint(0, 3)])
        p_pad = rnd.choice([None, len(p) + rnd.randint(0, 5)])
        for op in ("find", "rfind", "lt", "ge", "split", "rsplit", "split_inclusive", "split_terminator", "rsplit_terminator"):
            if p == "" and "split" in op:
                continue    # undefined in the reference (split.rs:337-338): covered by the table cases only
            bu = num_blocks_usize(255)
            ins = [string_blocks(a, a_pad), string_blocks(p, p_pad)]
            shape = [len(x) // 4 for x in ins]
            digits = np.concatenate(ins).astype("int64")
            ref = prog(op, shape, bu).run_plain(digits)
            fast = prog(op + "_logdepth", shape, bu).run_plain(digits)
            # message digits (mod 4; booleans mod 2 are covered by it) of every output block
            assert [int(v) % 4 for v in ref] == [int(v) % 4 for v in fast], (op, a, a_pad, p, p_pad)
    for trial in range(12):
        a = "".join(rnd.choice("xa") for _ in range(rnd.randint(1, 7)))
        p = "".join(rnd.choice("xa") for _ in range(rnd.randint(1, 2)))
        n = rnd.randint(0, 3)
        for op in ("splitn", "rsplitn"):   # truncate (split.rs:86-113) as a prefix count: every block of every slice
            assert run_ext(op + "_logdepth", [S(a), {"u": n}, S(p)])[0] == run_ext(op, [S(a), {"u": n}, S(p)])[0], (op, a, p, n)
            from fhe_string_amd import _lib as _l
            from tests.progs import build_inputs as _bi
            shape, digits = _bi(_l.load(), op, [S(a, len(a) + trial % 3), {"u": n}, S(p)], num_blocks_usize(255))
            ref = prog(op, shape, num_blocks_usize(255)).run_plain(digits.astype("int64"))
            fast = prog(op + "_logdepth", shape, num_blocks_usize(255)).run_plain(digits.astype("int64"))
            assert [int(v) % 4 for v in ref] == [int(v) % 4 for v in fast], (op, a, p, n)
        w = "".join(rnd.choice("a b\t") for _ in range(rnd.randint(0, 7)))
        for op in ("trim", "trim_start", "trim_end", "split_ascii_whitespace"):
            assert run_ext(op + "_logdepth", [S(w)])[0] == run_ext(op, [S(w)])[0], (op, w)
    # the barrel shifter behind substr_from (rotation modulo 4^Bu, zero-extended): every index incl. the
    # ones whose index + i wraps, every output block
    bu = num_blocks_usize(255)
    from fhe_string_amd import _lib
    from tests.progs import build_inputs
    for trial in range(40):
        la = rnd.randint(0, 9)
        a = "".join(rnd.choice("abc xy") for _ in range(la))
        pad = rnd.choice([None, la + rnd.randint(0, 3)])
        i, j = rnd.choice([rnd.randint(0, 12), rnd.randint(0, 255), 255, 250]), rnd.randint(0, 14)
        for op, args in (("substr_from", [S(a, pad), {"u": i}]), ("substr_end", [S(a, pad), {"u": i % 13}, {"u": j}]),
                         ("strip_prefix", [S(a, pad), S(a[:rnd.randint(0, 2)])]), ("split_once", [S(a, pad), S(rnd.choice(["a", "c ", "x"]))]),
                         ("rsplit_once", [S(a, pad), S(rnd.choice(["a", "c ", "x"]))]), ("strip_suffix", [S(a, pad), S(a[la - rnd.randint(0, 2):] if la else "")]),
                         # insert / add: char_at(b, i - index) and char_at(self, i - len(b)) as right rotations (round 3)
                         ("insert", [S(a, pad), {"u": rnd.choice([0, la, rnd.randint(0, 12), 255])}, S(rnd.choice(["", "Q", "QRS"]), rnd.choice([None, 4]))]),
                         ("add", [S(a, pad), S(rnd.choice(["", "Q", "QRS"]), rnd.choice([None, 4]))])):
            shape, digits = build_inputs(_lib.load(), op, args, bu)
            ref = prog(op, shape, bu).run_plain(digits.astype("int64"))
            fast = prog(op + "_logdepth", shape, bu).run_plain(digits.astype("int64"))
            assert [int(v) % 4 for v in ref] == [int(v) % 4 for v in fast], (op, a, pad, i, j)


def test_logdepth_depth_and_size(libs):
    """what the opt-in circuits buy at BASELINE config 3 / 5 sizes (levels = dependent launch pairs)"""
    bu = num_blocks_usize(512)
    for op, shape in (("split", [129, 4]), ("lt", [65, 65]), ("rfind", [129, 4])):
        ref, fast = prog(op, shape, bu), prog(op + "_logdepth", shape, bu)
        assert fast.n_levels * 4 < ref.n_levels, (op, fast.n_levels, ref.n_levels)
    ref, fast = prog("trim", [129], bu), prog("trim_logdepth", [129], bu)     # barrel shifter instead of slots^2 selections
    assert fast.n_pbs * 4 < ref.n_pbs and fast.n_levels * 2 < ref.n_levels, (fast.n_pbs, ref.n_pbs)


def test_results_cross_the_program_boundary_clean(libs):
    """every output block of every operation is handed over with the noise of ONE
    bootstrap at most and a degree below the message modulus -- the one-hot gathers (char_at,
    substr_from: sums of up to 257 bootstrapped blocks) included -- so chained API calls start from the
    operand the next circuit is sized for.  Under the multi-bit sets (noise cap 34) as well."""
    import ctypes as C
    import numpy as np
    from fhe_string_amd import _lib
    from fhe_string_amd.api import multibit_params
    lib = _lib.load()
    cases = [("find", [65, 5]), ("rfind", [33, 3]), ("len", [65]), ("char_at", [257]), ("substr_from", [41]), ("substr_to", [41]),
             ("trim", [17]), ("strip_prefix", [33, 3]), ("split", [17, 3]), ("split_logdepth", [17, 3]), ("replace", [9, 3, 2, 8]),
             ("to_uppercase", [9]), ("slices_reverse", [5]), ("ne", [9, 9]), ("insert", [9, 3]), ("find_logdepth", [65, 5])]
    for params in (default_params(), multibit_params(2), multibit_params(3)):
        for op, shape in cases:
            bu = num_blocks_usize(256 if shape[0] > 256 else 255)
            p = Prog(params, op, shape, bu)
            deg = np.zeros(p.n_outputs, dtype=np.int32)
            noise = np.zeros(p.n_outputs, dtype=np.float64)
            assert lib.fhes_program_out_meta(p.h, deg.ctypes.data, noise.ctypes.data) == 0
            assert (noise <= 1.0).all() and (noise >= 0).all(), (op, float(noise.max()))
            assert (deg < 4).all() and (deg >= 0).all(), (op, int(deg.max()))


def test_logdepth_replace_equals_the_references_loop(libs):
    """replace_logdepth (automaton + compaction network) against the faithful loop of replace.rs:65-99 on
    every output block: overlapping and padded patterns, empty pattern (the loop never advances and emits
    the replacement forever), empty replacement (the loop spends a slot on a zero), replacement longer
    and shorter than the pattern, outputs cut at l"""
    import numpy as np
    from fhe_string_amd import _lib
    from tests.progs import build_inputs
    rnd = random.Random(0x5EED)
    bu = num_blocks_usize(255)
    seen = {"empty_p": 0, "empty_s": 0, "grow": 0, "shrink": 0, "long_p": 0}
    for trial in range(190):
        la = rnd.randint(0, 8) if trial < 150 else rnd.randint(6, 11)
        a = "".join(rnd.choice("xa") for _ in range(la))
        p = "".join(rnd.choice("xa") for _ in range(rnd.randint(0, 3)))
        if trial >= 150:   # patterns of 4 .. 7 characters that DO occur (round 3: a length >= msg_mod was lost, see low_value)
            lp = rnd.randint(4, 7)
            i0 = rnd.randint(0, la - lp) if la >= lp else 0
            p = a[i0:i0 + lp] if la >= lp else "xaxa"
        seen["long_p"] += len(p) >= 4
        r = "".join(rnd.choice("yz") for _ in range(rnd.randint(0, 3)))
        a_pad = rnd.choice([None, la + rnd.randint(0, 3)])
        p_pad = rnd.choice([None, len(p) + rnd.randint(0, 3)])
        r_pad = rnd.choice([None, len(r) + rnd.randint(0, 1)])
        l = rnd.randint(1, 14)
        seen["empty_p"] += p == ""
        seen["empty_s"] += r == ""
        seen["grow"] += len(r) > len(p)
        seen["shrink"] += len(r) < len(p)
        args = [S(a, a_pad), S(p, p_pad), S(r, r_pad), {"i": l}]
        shape, digits = build_inputs(_lib.load(), "replace", args, bu)
        ref = prog("replace", shape, bu).run_plain(digits.astype("int64"))
        fast = prog("replace_logdepth", shape, bu).run_plain(digits.astype("int64"))
        assert [int(v) % 4 for v in ref] == [int(v) % 4 for v in fast], (a, a_pad, p, p_pad, r, r_pad, l)
    assert min(seen.values()) > 5, seen
    ref, fast = prog("replace", [65, 4, 2, 60], bu), prog("replace_logdepth", [65, 4, 2, 60], bu)
    assert fast.n_levels * 10 < ref.n_levels and fast.n_pbs * 2 < ref.n_pbs, (fast.n_pbs, ref.n_pbs, fast.n_levels, ref.n_levels)


def test_logdepth_replacen_equals_the_references_loop(libs):
    """replacen_logdepth: the first n_max occurrences of the unbounded greedy sequence (a prefix count masks the
    automaton's roles) against the faithful loop of replace.rs:65-99 with `n < n_max` on every output block -- n_max
    0, small, equal to and beyond the number of occurrences, empty pattern (n_max rounds of the replacement, then
    self from its start), empty replacement, padded operands, outputs cut at l"""
    from fhe_string_amd import _lib
    from tests.progs import build_inputs
    rnd = random.Random(0x5EED + 7)
    bu = num_blocks_usize(255)
    seen = {"empty_p": 0, "empty_s": 0, "n0": 0, "n_small": 0, "n_big": 0, "long_p": 0}
    for trial in range(240):
        la = rnd.randint(0, 8) if trial < 200 else rnd.randint(6, 11)
        a = "".join(rnd.choice("xa") for _ in range(la))
        p = "".join(rnd.choice("xa") for _ in range(rnd.randint(0, 3)))
        if trial >= 200:   # patterns of 4 .. 7 characters that occur
            lp = rnd.randint(4, 7)
            i0 = rnd.randint(0, la - lp) if la >= lp else 0
            p = a[i0:i0 + lp] if la >= lp else "xaxa"
        seen["long_p"] += len(p) >= 4
        r = "".join(rnd.choice("yz") for _ in range(rnd.randint(0, 3)))
        a_pad = rnd.choice([None, la + rnd.randint(0, 3)])
        p_pad = rnd.choice([None, len(p) + rnd.randint(0, 3)])
        r_pad = rnd.choice([None, len(r) + rnd.randint(0, 1)])
        l = rnd.randint(1, 14)
        n = rnd.choice([0, 1, 1, 2, 2, 3, 5, 200, 255])
        seen["empty_p"] += p == ""
        seen["empty_s"] += r == ""
        seen["n0"] += n == 0
        seen["n_small"] += 0 < n <= 3
        seen["n_big"] += n > 3
        args = [S(a, a_pad), S(p, p_pad), S(r, r_pad), {"u": n}, {"i": l}]
        shape, digits = build_inputs(_lib.load(), "replacen", args, bu)
        ref = prog("replacen", shape, bu).run_plain(digits.astype("int64"))
        fast = prog("replacen_logdepth", shape, bu).run_plain(digits.astype("int64"))
        assert [int(v) % 4 for v in ref] == [int(v) % 4 for v in fast], (a, a_pad, p, p_pad, r, r_pad, n, l)
    assert min(seen.values()) > 5, seen
    ref, fast = prog("replacen", [65, 4, 2, 60], bu), prog("replacen_logdepth", [65, 4, 2, 60], bu)
    assert fast.n_levels * 10 < ref.n_levels and fast.n_pbs * 2 < ref.n_pbs, (fast.n_pbs, ref.n_pbs, fast.n_levels, ref.n_levels)


def test_fuzzers_on_longer_inputs(libs):
    """a few trials of the two developer fuzzers (inputs up to 40 characters, occurring patterns of 1..7 characters):
    log-depth forms against serial forms, serial forms against Python's str where the reference behaves like std"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for script, seed, extra in (("scripts/fuzz_logdepth.py", "11", []), ("scripts/fuzz_std.py", "12", []),
                                ("scripts/fuzz_logdepth.py", "13", ["multibit2"])):   # the multi-bit sets: noise cap 34 instead of 256
        r = subprocess.run([sys.executable, script, "3", seed] + extra, cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_every_logdepth_form_compiles_under_the_multibit_noise_cap(libs):
    """round 3: with patterns of 4 and more characters the automaton's 3-bit states need a lookup on 8 * c + x -- 64 x the
    boolean's variance, more than the multi-bit sets' cap of 34 allows; those programs used to fail with "input noise
    exceeds the cap".  Every operation must compile for every pattern width under both multi-bit sets (wide patterns
    take the serial loop there)."""
    from fhe_string_amd.api import multibit_params
    bu = num_blocks_usize(255)
    for g in (2, 3):
        params = multibit_params(g)
        for op in ("replace", "replacen", "split", "rsplit", "splitn", "split_inclusive", "split_once", "strip_suffix", "find"):
            for p_slots in (2, 4, 5, 6, 8, 10):
                shape = [21, p_slots, 3, 20] if op.startswith("replace") else [21, p_slots]
                Prog(params, op + "_logdepth", shape, bu)


def test_find_pred_scans(libs):
    """find_pred_unchecked / rfind_pred_unchecked (search.rs:196-213): the index scan over the closure's results"""
    rnd = random.Random(11)
    for op in ("find_pred", "rfind_pred", "find_pred_logdepth", "rfind_pred_logdepth"):
        for n in (0, 1, 2, 7, 33):
            for _ in range(4):
                v = [rnd.randint(0, 1) if rnd.random() < 0.8 else 0 for _ in range(n)]
                hits = [i for i, x in enumerate(v) if x]
                want = (hits[-1] if op.startswith("r") else hits[0]) if hits else None
                assert run_ext(op, [{"l": v}])[0] == want, (op, v)


def _staged_split_plain(op, a, a_pad, p, p_pad, cut, suffix=""):
    """the position-sharded form of split / rsplit (csrc/comm.hip: fhes_sharded_op) run stage by stage on the
    plaintext model: find_all over two position ranges, head, is_some over two index ranges, tail"""
    import numpy as np
    bu = num_blocks_usize(255)
    A, P = string_blocks(a, a_pad), string_blocks(p, p_pad)
    sa, sp = len(A) // 4, len(P) // 4
    n_pos, n = sa - 1, sa + 1
    ap = np.concatenate([A, P]).astype("int64")
    c1 = min(cut, n_pos)
    found = np.concatenate([prog("find_all_range", [sa, sp, lo, hi], bu).run_plain(ap) if hi > lo else np.zeros(0, dtype="int64")
                            for lo, hi in ((0, c1), (c1, n_pos))])
    head = prog(op + "_head" + suffix, [sa, sp], bu).run_plain(np.concatenate([ap, found]))
    matches, p_len, s_len, p_empty = head[:n], head[n:n + bu], head[n + bu:n + 2 * bu], head[n + 2 * bu:]
    c2 = min(cut + 1, n)
    is_some = np.concatenate([prog("split_is_some_range" + suffix, [n, lo, hi], bu).run_plain(np.concatenate([matches, p_len, s_len]))
                              for lo, hi in ((0, c2), (c2, n))])
    return prog(op + "_tail" + suffix, [n], bu).run_plain(np.concatenate([matches, is_some, p_len, s_len, p_empty])), bu


def test_staged_split_equals_the_monolithic_program(libs):
    """every output block of the staged (shardable) split / rsplit equals the monolithic program's, faithful and
    log-depth forms, on the reference's split cases and on random inputs; decoded: the std::str answer"""
    rnd = random.Random(5)
    cases = [("xxx", None, "x", None), ("axbxc", None, "x", None), ("xxx", None, "xx", None), ("abc", None, "", None),
             ("axbxc", 8, "x", 3), ("xxx", 8, "xx", None), ("", None, "x", None), ("a", 3, "", 2)]
    for _ in range(10):
        a = "".join(rnd.choice("ab,") for _ in range(rnd.randint(0, 9)))
        p = "".join(rnd.choice("ab,") for _ in range(rnd.randint(0, 2)))
        cases.append((a, rnd.choice([None, len(a) + 2]), p, rnd.choice([None, len(p) + 1])))
    for a, a_pad, p, p_pad in cases:
        for op in ("split", "rsplit"):
            for suffix in ("", "_logdepth"):
                mono, _ = run_plain(op + suffix, a, a_pad, p, p_pad)
                import numpy as np
                bu = num_blocks_usize(255)
                mp = prog(op + suffix, [len(string_blocks(a, a_pad)) // 4, len(string_blocks(p, p_pad)) // 4], bu)
                mono_raw = mp.run_plain(np.concatenate([string_blocks(a, a_pad), string_blocks(p, p_pad)]).astype("int64"))
                for cut in (0, 2, 100):
                    raw, _ = _staged_split_plain(op, a, a_pad, p, p_pad, cut, suffix)
                    assert list(raw % 4) == list(mono_raw % 4), (a, a_pad, p, p_pad, op, suffix, cut)
                    assert decode_result(op, raw, bu, a) == mono, (a, p, op, suffix, cut)


def test_rest_programs_equal_the_monolithic_programs(libs):
    """`"<op>.rest"` -- the operation with its find_all(self, pattern) handed in (what `fhes_sharded_op_ex` runs replicated
    behind the sharded find_all) -- against the operation itself, every output block, faithful and log-depth forms"""
    import numpy as np
    bu = num_blocks_usize(255)
    rnd = random.Random(9)
    ops = [("rfind", []), ("find", []), ("contains", []), ("split", []), ("rsplit", []), ("split_inclusive", []), ("split_terminator", []),
           ("rsplit_terminator", []), ("split_once", []), ("rsplit_once", []), ("strip_suffix", []), ("replace", ["sub", "l"]),
           ("replacen", ["sub", "u", "l"]), ("splitn", ["u_mid"]), ("rsplitn", ["u_mid"])]
    for trial in range(6):
        a = "".join(rnd.choice("abx") for _ in range(rnd.randint(1, 8)))
        pt = "".join(rnd.choice("abx") for _ in range(rnd.randint(0, 2)))
        a_pad, p_pad = rnd.choice([None, len(a) + 2]), rnd.choice([None, len(pt) + 1])
        A, P, SUB = string_blocks(a, a_pad), string_blocks(pt, p_pad), string_blocks("Z")
        U = np.array([2, 0, 0, 0], dtype=np.int64)
        sa, sp = len(A) // 4, len(P) // 4
        found = prog("find_all_range", [sa, sp, 0, sa - 1], bu).run_plain(np.concatenate([A, P]).astype("int64"))
        for op, extra in ops:
            if extra == ["u_mid"]:          # signature SUS: (self, n, pattern)
                shape, ins = [sa, sp], [A, U, P]
            else:
                shape, ins = [sa, sp], [A, P]
                for x in extra:
                    if x == "sub":
                        shape.append(len(SUB) // 4)
                        ins.append(SUB)
                    elif x == "u":
                        ins.append(U)
                    else:
                        shape.append(len(a) + 4)
            for suffix in ("", "_logdepth"):
                mono = prog(op + suffix, shape, bu).run_plain(np.concatenate(ins).astype("int64"))
                rest = prog(op + ".rest" + suffix, shape, bu).run_plain(np.concatenate(ins + [found]).astype("int64"))
                assert list(mono % 4) == list(rest % 4), (a, a_pad, pt, p_pad, op, suffix)
    with pytest.raises(Exception):
        prog("len.rest", [4], bu)       # no (self, pattern): nothing to hand in


def test_crate_public_helpers_of_logic_rs_and_mod_rs(libs):
    """`any` / `all` (logic.rs:41-65), `element_at_bool` (mod.rs:271-287), `if_then_else_zero`, `scalar_if_then_else_zero`,
    `if_then_else_bool` (logic.rs:13-38) as operations of their own (the Rust binding exposes them like the reference's crate does)"""
    import numpy as np
    rng = random.Random(31)
    bu = num_blocks_usize(255)

    def radix(v):
        return [(v >> (2 * k)) & 3 for k in range(bu)]

    def val(blocks):
        return sum(int(b) << (2 * k) for k, b in enumerate(blocks))

    for n in (0, 1, 2, 7, 16, 40):
        for _ in range(4):
            flags = [rng.randint(0, 1) for _ in range(n)]
            for op, want in (("any", int(any(flags))), ("all", int(all(flags)))):
                p = prog(op, [n], bu)
                assert [int(v) for v in p.run_plain(np.array(flags, dtype="int64"))] == [want], (op, flags)
            for i in {0, max(n - 1, 0), n, n + 3, rng.randint(0, 255)}:
                p = prog("element_at_bool", [n], bu)
                out = p.run_plain(np.array(flags + radix(i), dtype="int64"))
                assert [int(v) for v in out] == [flags[i] if i < n else 0], (flags, i)
    for _ in range(20):
        a, x, c = rng.randint(0, 1), rng.randint(0, 255), rng.randint(0, 255)
        assert val(prog("if_then_else_zero", [], bu).run_plain(np.array([a] + radix(x), dtype="int64"))) == (x if a else 0)
        assert val(prog("scalar_if_then_else_zero", [c], bu).run_plain(np.array([a], dtype="int64"))) == (c if a else 0)
        b1, b2 = rng.randint(0, 1), rng.randint(0, 1)
        assert [int(v) for v in prog("if_then_else_bool", [], bu).run_plain(np.array([a, b1, b2], dtype="int64"))] == [b1 if a else b2]


def test_wide_usize_paths_of_round_4(libs):
    """five and six usize blocks (l = 1023 / 4095: THREE packed pairs per comparison, so `reduce_signs` takes its three-signs-per-lookup
    form, and the one-digit additions chain their carries through more than four blocks): the serial loops that live on counters and
    comparisons -- split / rsplit / splitn, replace / replacen, find / rfind, len, lt -- against Python, faithful and opt-in forms"""
    rnd = random.Random(0xF4E54)
    for l in (1023, 4095):
        for trial in range(10):
            la, lb = rnd.randint(1, 9), rnd.randint(1, 3)
            a = "".join(rnd.choice("ab,") for _ in range(la))
            b = a[:lb] if trial % 2 else "".join(rnd.choice("ab,") for _ in range(lb))
            a_pad, b_pad = rnd.choice([None, la + 2]), rnd.choice([None, lb + 1])
            ctx = (l, a, a_pad, b, b_pad)
            for form in ("", "_logdepth"):
                assert run_plain("split" + form, a, a_pad, b, b_pad, l=l)[0] == rust_split(a, b), ctx
                # (rsplit of a padded multi-character pattern follows the reference's own loop, not Python: the two forms agree with each other)
                assert run_plain("rsplit" + form, a, a_pad, b, b_pad, l=l)[0] == run_plain("rsplit", a, a_pad, b, b_pad, l=255)[0], ctx
                assert run_plain("find" + form, a, a_pad, b, b_pad, l=l)[0] == rust_find(a, b), ctx
                assert run_plain("rfind" + form, a, a_pad, b, b_pad, l=l)[0] == rust_rfind(a, b), ctx
                assert run_plain("lt" + form, a, a_pad, b, b_pad, l=l)[0] == (a < b), ctx
            assert run_plain("len", a, a_pad, l=l)[0] == len(a), ctx
            got = run_ext("replace", [S(a, a_pad), S(b, b_pad), S("xy", None), {"i": la + 6}], l=l)[0]
            assert got == a.replace(b, "xy")[:la + 6], ctx
            n = rnd.randint(0, 3)
            got = run_ext("replacen", [S(a, a_pad), S(b, b_pad), S("z", None), {"u": n}, {"i": la + 4}], l=l)[0]
            assert got == a.replace(b, "z", n)[:la + 4], ctx
            got = run_ext("splitn", [S(a, a_pad), {"u": n}, S(b, b_pad)], l=l)[0]
            assert got == (a.split(b, n - 1) if n > 0 else []), ctx
