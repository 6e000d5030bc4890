"""Writes tests/golden/reference_tables.json: the plaintext-level pins of the reference.

The reference's tests assert decrypt(fhe_op(encrypt(x))) == std::str op(x)
(/root/reference/src/ciphertext/tests/*.rs).  Inputs below are those tables (file:line
cited per group); expected values are Rust `std::str` results, written out literally
where SURVEY.md section 8c lists them and otherwise computed by `rust_*` helpers that
restate the std semantics (the traps: "abc".split("") == ["", "a", "b", "c", ""];
rfind("") == len).
"""
import json
import os


def rust_split(s, p):
    if p == "":
        return [""] + list(s) + [""]
    return s.split(p)


def rust_rsplit(s, p):
    # Rust searches from the right (non-overlapping matches are chosen right to left)
    if p == "":
        return list(reversed(rust_split(s, p)))
    return list(reversed(s.rsplit(p)))


def rust_split_inclusive(s, p):
    parts, out = s.split(p), []
    for i, x in enumerate(parts):
        if i < len(parts) - 1:
            out.append(x + p)
        elif x != "":
            out.append(x)
    return out


def rust_split_terminator(s, p):
    v = rust_split(s, p)
    if v and v[-1] == "":
        v = v[:-1]
    return v


def rust_rsplit_terminator(s, p):
    # back-to-front iteration; the piece after the last match is skipped if empty
    v = rust_rsplit(s, p)
    if v and v[0] == "":
        v = v[1:]
    return v


def opt(v):
    return None if v < 0 else v


cases = []


def add(op, a, a_pad, b=None, b_pad=None, expect=None, src=""):
    cases.append({"op": op, "a": a, "a_pad": a_pad, "b": b, "b_pad": b_pad, "expect": expect, "src": src})


# tests/search.rs:15-40 (find / rfind / contains), :96-115 (starts_with / ends_with)
search = [("abcabc", None, "ab", None), ("abc", 3, "d", 5), ("abcdef", None, "cde", None), ("abc", None, "", None)]
find_expect = [0, None, 2, 0]
rfind_expect = [3, None, 2, 3]
contains_expect = [True, False, True, True]
for (a, ap, b, bp), f, r, c in zip(search, find_expect, rfind_expect, contains_expect):
    assert opt(a.find(b)) == f and opt(a.rfind(b)) == r and (b in a) == c
    add("find", a, ap, b, bp, f, "tests/search.rs:15-40")
    add("rfind", a, ap, b, bp, r, "tests/search.rs:15-40")
    add("contains", a, ap, b, bp, c, "tests/search.rs:15-40")
for (a, ap, b, bp), sw, ew in zip(search[:3], [True, False, False], [False, False, False]):
    assert a.startswith(b) == sw and a.endswith(b) == ew
    add("starts_with", a, ap, b, bp, sw, "tests/search.rs:96-115")
    add("ends_with", a, ap, b, bp, ew, "tests/search.rs:96-115")

# tests/compare.rs:13-38 (is_empty)
for a, ap in [("a", None), ("a", 3), ("abc", None), ("abc", 5), ("", None), ("", 3)]:
    add("is_empty", a, ap, expect=(a == ""), src="tests/compare.rs:13-38")
# tests/compare.rs:67-110 (eq / ne): T,T,F,F,F,F,T
eqs = [("abc", None, "abc", None), ("abc", 3, "abc", 5), ("abc", None, "abcd", None), ("abcd", None, "ab", None),
       ("a", None, "b", None), ("a", None, "", None), ("", None, "", None)]
for (a, ap, b, bp), e in zip(eqs, [True, True, False, False, False, False, True]):
    assert (a == b) == e
    add("eq", a, ap, b, bp, e, "tests/compare.rs:67-110")
    add("ne", a, ap, b, bp, not e, "tests/compare.rs:67-110")
# tests/compare.rs:156-199 (le, lt, ge, gt): TFTF, TFTF, TTFF, FFTT, TTFF, FFTT, FFTT
cmps = [("abc", None, "abc", None), ("abc", 3, "abc", 5), ("abc", None, "abcd", None), ("abcd", None, "ab", None),
        ("abc", None, "bcd", None), ("cde", None, "abc", None), ("a", None, "", None)]
rows = ["TFTF", "TFTF", "TTFF", "FFTT", "TTFF", "FFTT", "FFTT"]
for (a, ap, b, bp), row in zip(cmps, rows):
    for op, ch, py in zip(["le", "lt", "ge", "gt"], row, [a <= b, a < b, a >= b, a > b]):
        assert (ch == "T") == py
        add(op, a, ap, b, bp, ch == "T", "tests/compare.rs:156-199")
# tests/mod.rs:47-60 (len)
for a, ap in [("", None), ("abc", None), ("abc", 8)]:
    add("len", a, ap, expect=len(a), src="tests/mod.rs:47-60")
# tests/split.rs:17-48 (split); pad applies to input AND pattern (split.rs:51-52)
splits = [("xxx", "x", None), ("axbxc", "x", None), ("xxx", "xx", None), ("abc", "", None), ("axbxc", "x", 8), ("abc", "", 8)]
split_expect = [["", "", "", ""], ["a", "b", "c"], ["", "x"], ["", "a", "b", "c", ""], ["a", "b", "c"], ["", "a", "b", "c", ""]]
for (a, p, pad), e in zip(splits, split_expect):
    assert rust_split(a, p) == e
    add("split", a, pad, p, pad, e, "tests/split.rs:17-48")
# tests/split.rs:137-155 (split_terminator), :188-206 (split_inclusive), :284-317 (rsplit), :350-368 (rsplit_terminator)
four = [("xxx", "x", None), ("axbxc", "x", None), ("xxx", "xx", None), ("axbxc", "x", 8)]
for a, p, pad in four:
    add("split_terminator", a, pad, p, pad, rust_split_terminator(a, p), "tests/split.rs:137-155")
    add("split_inclusive", a, pad, p, pad, rust_split_inclusive(a, p), "tests/split.rs:188-206")
    add("rsplit_terminator", a, pad, p, pad, rust_rsplit_terminator(a, p), "tests/split.rs:350-368")
for a, p, pad in [("xxx", "x", None), ("axbxc", "x", None), ("xxx", "xx", None), ("xaxbx", "x", None), ("axb", "x", None),
                  ("abxxcdxxef", "xx", None), ("axbxc", "x", 8)]:
    add("rsplit", a, pad, p, pad, rust_rsplit(a, p), "tests/split.rs:284-317")
# README.md:34 / examples/cmd: input " A bcbc ", pattern "bc" (BASELINE config 1)
a, b = " A bcbc ", "bc"
add("contains", a, None, b, None, True, "README.md:34")
add("find", a, None, b, None, 3, "README.md:34")
add("rfind", a, None, b, None, 5, "README.md:34")
add("eq", a, None, b, None, False, "README.md:34")
add("split", a, None, b, None, [" A ", "", " "], "README.md:34")

# ---- second table: operations with FheUsize / clear-int operands or string results ----------
# generic form: {"op", "args": [operand...], "expect", "src"}; operand = {"s": str, "pad": int|None}
# for an FheString, {"u": int} for an encrypted FheUsize, {"i": int} for a clear integer; operand
# order as in the reference's method signature (self first).
ext = []


def S(s, pad=None):
    return {"s": s, "pad": pad}


def addx(op, args, expect, src):
    ext.append({"op": op, "args": args, "expect": expect, "src": src})


WS = " \t\n\x0b\x0c\r"
# tests/trim.rs:15-36
for a, pad in [("   abc   ", None), ("abc   ", None), ("   abc", None), ("abc", None), ("a ", 3), ("", None)]:
    addx("trim", [S(a, pad)], a.strip(WS), "tests/trim.rs:15-36")
    addx("trim_start", [S(a, pad)], a.lstrip(WS), "tests/trim.rs:15-36")
    addx("trim_end", [S(a, pad)], a.rstrip(WS), "tests/trim.rs:15-36")
# tests/trim.rs:95-116
for a, b in [("abc", "ab"), ("abc", "bc"), ("abc", "abc"), ("abc", "def")]:
    addx("strip_prefix", [S(a), S(b)], a[len(b):] if a.startswith(b) else None, "tests/trim.rs:95-116")
    addx("strip_suffix", [S(a), S(b)], a[:len(a) - len(b)] if a.endswith(b) else None, "tests/trim.rs:95-116")
# tests/replace.rs:20-41 (pad applies to input, pattern and replacement; l = result length)
for a, p, r, pad in [("aa", "a", "b", None), ("abdb", "b", "c", None), ("aa", "aa", "b", None), ("ababcd", "ab", "c", 8)]:
    res = a.replace(p, r)
    addx("replace", [S(a, pad), S(p, pad), S(r, pad), {"i": len(res)}], res, "tests/replace.rs:20-41")
# tests/replace.rs:78-96
for a, p, r, pad, n in [("aaa", "a", "b", None, 2), ("abdb", "b", "c", 4, 1), ("aaaa", "aa", "b", None, 3)]:
    res = a.replace(p, r, n)
    addx("replacen", [S(a, pad), S(p, pad), S(r, pad), {"u": n}, {"i": len(res)}], res, "tests/replace.rs:78-96")
# tests/insert.rs:22-43 (add)
for a, b in [("abc", "def"), ("abc", ""), ("", "def"), ("", "")]:
    addx("add", [S(a), S(b)], a + b, "tests/insert.rs:22-43")
# tests/insert.rs:64-112 (repeat): operands (self, n, l)
for a, n, l in [("abc", 0, 8), ("abc", 1, 8), ("abc", 2, 8)]:
    addx("repeat", [S(a), {"u": n}, {"i": l}], a * n, "tests/insert.rs:64-112")
# tests/convert.rs:15-24
for a in ["ABC", "Abc", "abc def"]:
    addx("to_lowercase", [S(a)], a.lower(), "tests/convert.rs:15-24")
    addx("to_uppercase", [S(a)], a.upper(), "tests/convert.rs:15-24")
# tests/compare.rs:271-290
for a, ap, b, bp in [("abc", None, "ABC", None), ("AbC", 3, "aBc", 5), ("ABC", None, "def", None)]:
    addx("eq_ignore_ascii_case", [S(a, ap), S(b, bp)], a.lower() == b.lower(), "tests/compare.rs:271-290")
# tests/split.rs:81-102 (splitn), :402-423 (rsplitn): operands (self, n, pattern)
for a, p, pad, n in [("xxx", "x", None, 1), ("axbxc", "x", None, 2), ("xxx", "xx", None, 3), ("axbxc", "x", 8, 2)]:
    addx("splitn", [S(a, pad), {"u": n}, S(p, pad)], a.split(p, n - 1) if n else [], "tests/split.rs:81-102")
    addx("rsplitn", [S(a, pad), {"u": n}, S(p, pad)], list(reversed(a.rsplit(p, n - 1))) if n else [], "tests/split.rs:402-423")
# tests/split.rs:238-251
for a, pad in [(" x x x ", None), ("ab cd ed", None), ("ab", None), (" x x x ", 8)]:
    addx("split_ascii_whitespace", [S(a, pad)], a.split(), "tests/split.rs:238-251")
# tests/split.rs:458-477 (split_once), :518-537 (rsplit_once)
for a, p, pad in [("xxx", "x", None), ("axbxc", "x", None), ("xxx", "xx", None), ("axbxc", "x", 8)]:
    i = a.find(p)
    addx("split_once", [S(a, pad), S(p, pad)], [a[:i], a[i + len(p):]] if i >= 0 else None, "tests/split.rs:458-477")
    i = a.rfind(p)
    addx("rsplit_once", [S(a, pad), S(p, pad)], [a[:i], a[i + len(p):]] if i >= 0 else None, "tests/split.rs:518-537")

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_tables.json")
with open(out, "w") as f:
    json.dump({"source": "matthiasgeihs/fhe_string src/ciphertext/tests/*.rs, README.md", "cases": cases,
               "cases_ext": ext}, f, indent=1)
print(len(cases), "+", len(ext), "cases ->", out)
