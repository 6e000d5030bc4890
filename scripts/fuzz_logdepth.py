"""Developer fuzz (CPU, plaintext model of the compiled circuits): every operation's log-depth form against its serial
form on random inputs that are LONGER than the unit tests' (inputs up to 40 characters, occurring patterns of 1..7
characters, paddings, n_max).  Prints every disagreement; exit code 1 if there is one.
    python scripts/fuzz_logdepth.py [trials] [seed] [multibit2|multibit3]"""
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

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0


def both(op, args, ctx):
    global bad
    a, b = run_ext(op, args)[0], run_ext(op + "_logdepth", args)[0]
    if a != b:
        bad += 1
        print("MISMATCH", op, ctx, "serial:", a, "log-depth:", b, flush=True)


for t in range(trials):
    la = rnd.randint(8, 40)
    alpha = rnd.choice(["ab", "abc", "ab ", "xy,"])
    a = "".join(rnd.choice(alpha) for _ in range(la))
    lp = rnd.randint(1, 7)
    i0 = rnd.randint(0, la - lp)
    p = a[i0:i0 + lp] if rnd.random() < 0.8 else "".join(rnd.choice(alpha) for _ in range(lp))
    a_pad = rnd.choice([None, la + rnd.randint(0, 4)])
    p_pad = rnd.choice([None, min(lp + rnd.randint(0, 3), 8)])
    r = "".join(rnd.choice("QR") for _ in range(rnd.randint(0, 4)))
    n = rnd.choice([0, 1, 2, 3, 7, 255])
    ctx = (a, a_pad, p, p_pad, r, n)
    A, P, R = S(a, a_pad), S(p, p_pad), S(r)
    for op in ("find", "rfind", "split", "rsplit", "split_inclusive", "split_terminator", "rsplit_terminator", "split_once",
               "rsplit_once", "strip_prefix", "strip_suffix", "lt", "ge"):
        both(op, [A, P], ctx)
    for op in ("splitn", "rsplitn"):
        both(op, [A, {"u": n}, P], ctx)
    l = rnd.randint(1, la + 8)
    both("replace", [A, P, R, {"i": l}], ctx + (l,))
    both("replacen", [A, P, R, {"u": n}, {"i": l}], ctx + (l,))
    both("insert", [A, {"u": rnd.randint(0, la + 2)}, S(r, rnd.choice([None, 5]))], ctx)
    both("add", [A, S(r, rnd.choice([None, 5]))], ctx)
    i, j = rnd.randint(0, la + 3), rnd.randint(0, la + 3)
    both("substr_end", [A, {"u": i}, {"u": j}], ctx + (i, j))
    both("substr_from", [A, {"u": i}], ctx + (i,))
    w = "".join(rnd.choice("a b\t") for _ in range(rnd.randint(0, 30)))
    for op in ("trim", "trim_start", "trim_end", "split_ascii_whitespace"):
        both(op, [S(w, rnd.choice([None, len(w) + 2]))], (w,))
    print(f"trial {t + 1}/{trials}: {bad} mismatch(es) so far", flush=True)
sys.exit(1 if bad else 0)
