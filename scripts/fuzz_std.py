"""Developer fuzz (CPU, plaintext model): the SERIAL circuits (the default) against Python's str on inputs longer than the
unit tests' -- only where the reference's behaviour is std's (non-empty occurring or random patterns, no empty-pattern
split / replace).  Prints every disagreement; exit code 1 if there is one.
    python scripts/fuzz_std.py [trials] [seed] [multibit2|multibit3]"""
import random
import sys

sys.path.insert(0, ".")
import tests.test_circuits_plain as tcp  # noqa: E402
from tests.test_circuits_plain import S, run_ext  # noqa: E402

if len(sys.argv) > 3:   # parameter set: "multibit2" / "multibit3" compile the same circuits under a noise cap of 34 instead of 256
    import ctypes as C
    from fhe_string_amd._lib import Params, check, load
    _p = Params()
    check(load(strict=False).fhes_params_multibit(C.byref(_p), int(sys.argv[3][-1])))
    tcp.default_params = lambda: _p
    tcp._cache.clear()

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0


def check(op, args, want, ctx):
    global bad
    got = run_ext(op, args)[0]
    if got != want:
        bad += 1
        print("MISMATCH", op, ctx, "circuit:", got, "std:", want, flush=True)


for t in range(trials):
    la = rnd.randint(6, 32)
    alpha = rnd.choice(["ab", "abc", "ab ", "xy,"])
    a = "".join(rnd.choice(alpha) for _ in range(la))
    lp = rnd.randint(1, 6)
    i0 = rnd.randint(0, la - lp)
    p = a[i0:i0 + lp] if rnd.random() < 0.8 else "".join(rnd.choice(alpha) for _ in range(lp))
    a_pad = rnd.choice([None, la + rnd.randint(0, 3)])
    p_pad = rnd.choice([None, lp + rnd.randint(0, 2)])
    r = "".join(rnd.choice("QR") for _ in range(rnd.randint(1, 3)))
    n = rnd.choice([0, 1, 2, 5])
    ctx = (a, a_pad, p, p_pad, r, n)
    A, P, R = S(a, a_pad), S(p, p_pad), S(r)
    i = a.find(p)
    check("find", [A, P], None if i < 0 else i, ctx)
    i = a.rfind(p)
    check("rfind", [A, P], None if i < 0 else i, ctx)
    check("contains", [A, P], p in a, ctx)
    check("starts_with", [A, P], a.startswith(p), ctx)
    check("split", [A, P], a.split(p), ctx)
    # rsplit / rsplitn with a pattern of more than one character follow the REFERENCE, not std: its
    # rfind_all_non_overlapping drops every match at a position k with k + 1 < len(p) and never suppresses overlaps
    # (search.rs:136-149; tests/ref_model.py restates it, tests/test_circuits_plain.py pins it)
    if len(p) == 1:
        check("rsplit", [A, P], list(reversed(a.rsplit(p))), ctx)
        check("rsplitn", [A, {"u": n}, P], [] if n == 0 else list(reversed(a.rsplit(p, n - 1))), ctx)
    check("splitn", [A, {"u": n}, P], [] if n == 0 else a.split(p, n - 1), ctx)
    rep = a.replace(p, r)
    check("replace", [A, P, R, {"i": len(rep) + 2}], rep, ctx)
    repn = a.replace(p, r, n)
    check("replacen", [A, P, R, {"u": n}, {"i": len(repn) + 2}], repn, ctx)
    check("lt", [A, P], a < p, ctx)
    check("ge", [A, P], a >= p, ctx)
    check("eq", [A, S(a, a_pad)], True, ctx)
    check("len", [A], len(a), ctx)
    w = "".join(rnd.choice("a b\t") for _ in range(rnd.randint(0, 24)))
    check("trim", [S(w, rnd.choice([None, len(w) + 2]))], w.strip(" \t"), (w,))
    check("split_ascii_whitespace", [S(w)], w.split(), (w,))
    k = rnd.randint(0, la)
    check("insert", [A, {"u": k}, R], a[:k] + r + a[k:], ctx + (k,))
    check("add", [A, R], a + r, ctx)
    i = a.find(p)
    check("split_once", [A, P], [a[:i], a[i + len(p):]] if i >= 0 else None, ctx)
    j = a.rfind(p)
    check("rsplit_once", [A, P], [a[:j], a[j + len(p):]] if j >= 0 else None, ctx)
    check("strip_prefix", [A, P], a[len(p):] if a.startswith(p) else None, ctx)
    parts = a.split(p)
    check("split_inclusive", [A, P], [x + p for x in parts[:-1]] + ([parts[-1]] if parts[-1] else []), ctx)
    check("split_terminator", [A, P], parts[:-1] if parts[-1] == "" else parts, ctx)
    x, y = sorted((rnd.randint(0, la), rnd.randint(0, la)))
    check("substr_from", [A, {"u": x}], a[x:], ctx + (x,))
    check("substr_to", [A, {"u": y}], a[:y], ctx + (y,))
    check("substr_end", [A, {"u": x}, {"u": y}], a[x:y], ctx + (x, y))
    m = rnd.randint(0, 3)
    check("repeat", [P, {"u": m}, {"i": 3 * len(p) + 1}], (p * m)[:3 * len(p) + 1], ctx + (m,))
    check("to_uppercase", [A], a.upper(), ctx)
    check("trim_start", [S(w)], w.lstrip(" \t"), (w,))
    check("trim_end", [S(w)], w.rstrip(" \t"), (w,))
    print(f"trial {t + 1}/{trials}: {bad} mismatch(es) so far", flush=True)
sys.exit(1 if bad else 0)
