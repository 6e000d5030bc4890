"""Developer fuzz on a GPU box: random operations of the Level-2 API at the REAL parameter sets on strings of 10-30
characters with occurring patterns of 1-6 characters, serial and log-depth circuits, against Python's str (only where the
reference behaves like std).  Prints a progress line per trial (the pool's silence watchdog) and every disagreement.
    python scripts/gpu_fuzz.py [trials] [seed] [classic|multibit2|multibit3]"""
import random
import sys
import time

sys.path.insert(0, ".")
from fhe_string_amd import api  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 6
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
which = sys.argv[3] if len(sys.argv) > 3 else "classic"
params = api.default_params() if which == "classic" else api.multibit_params(int(which[-1]))
ck, sk = api.generate_keys_with_params(params, 255, seed=0xF022 + seed)
rnd = random.Random(seed)
E = lambda s, pad=None: api.FheString.new(ck, s, pad)  # noqa: E731
U = lambda v: api.FheUsize.new(ck, v)  # noqa: E731
bad = 0


def check(name, got, want, ctx):
    global bad
    if got != want:
        bad += 1
        print("MISMATCH", name, ctx, "got:", got, "want:", want, flush=True)


t_start = time.time()
for t in range(trials):
    la = rnd.randint(10, 30)
    a = "".join(rnd.choice("ab ,") for _ in range(la))
    lp = rnd.randint(1, 6)
    i0 = rnd.randint(0, la - lp)
    p = a[i0:i0 + lp] if rnd.random() < 0.8 else "".join(rnd.choice("ab,") for _ in range(lp))
    pad, p_pad = rnd.choice([None, la + 2]), rnd.choice([None, lp + 1])
    n = rnd.randint(0, 3)
    r = rnd.choice(["y", "zz"])
    A, P = E(a, pad), E(p, p_pad)
    for mode in ("serial", "logdepth"):
        sk.set_logdepth(mode == "logdepth")
        ctx = (mode, a, pad, p, p_pad)
        i = a.find(p)
        check("find", A.find(sk, P).decrypt(ck), None if i < 0 else i, ctx)
        j = a.rfind(p)
        check("rfind", A.rfind(sk, P).decrypt(ck), None if j < 0 else j, ctx)
        check("contains", ck.decrypt_bool(A.contains(sk, P)), p in a, ctx)
        check("split", A.split(sk, P).decrypt(ck), a.split(p), ctx)
        check("splitn", A.splitn(sk, U(n), P).decrypt(ck), a.split(p, n - 1) if n else [], ctx + (n,))
        res, resn = a.replace(p, r), a.replace(p, r, n)
        check("replace", A.replace(sk, P, E(r), len(res) + 1).decrypt(ck), res, ctx + (r,))
        check("replacen", A.replacen(sk, P, E(r), U(n), len(resn) + 1).decrypt(ck), resn, ctx + (r, n))
        check("trim", A.trim(sk).decrypt(ck), a.strip(" "), ctx)
        k = rnd.randint(0, la)
        check("insert", A.insert(sk, U(k), E("QR")).decrypt(ck), a[:k] + "QR" + a[k:], ctx + (k,))
        check("lt", ck.decrypt_bool(A.lt(sk, P)), a < p, ctx)
        check("len", A.len(sk).decrypt(ck), la, ctx)
        if t % 2 == 1:   # every other trial: the rest of the API whose behaviour is std's
            B = ck.decrypt_bool
            check("le/gt/ge", (B(A.le(sk, P)), B(A.gt(sk, P)), B(A.ge(sk, P))), (a <= p, a > p, a >= p), ctx)
            check("eq/ne", (B(A.eq(sk, E(a, la + 1))), B(A.ne(sk, P))), (True, a != p), ctx)
            check("starts_with", B(A.starts_with(sk, P)), a.startswith(p), ctx)
            i = a.find(p)
            check("split_once", A.split_once(sk, P).decrypt(ck), (a[:i], a[i + len(p):]) if i >= 0 else None, ctx)
            j = a.rfind(p)
            check("rsplit_once", A.rsplit_once(sk, P).decrypt(ck), (a[:j], a[j + len(p):]) if j >= 0 else None, ctx)
            check("strip_prefix", A.strip_prefix(sk, P).decrypt(ck), a[len(p):] if a.startswith(p) else None, ctx)
            parts = a.split(p)
            check("split_inclusive", A.split_inclusive(sk, P).decrypt(ck),
                  [x + p for x in parts[:-1]] + ([parts[-1]] if parts[-1] else []), ctx)
            check("split_terminator", A.split_terminator(sk, P).decrypt(ck), parts[:-1] if parts[-1] == "" else parts, ctx)
            check("split_ascii_whitespace", A.split_ascii_whitespace(sk).decrypt(ck), a.split(), ctx)
            check("trim_start/end", (A.trim_start(sk).decrypt(ck), A.trim_end(sk).decrypt(ck)), (a.lstrip(" "), a.rstrip(" ")), ctx)
            check("upper/lower", (A.to_uppercase(sk).decrypt(ck), E(a.upper(), pad).to_lowercase(sk).decrypt(ck)), (a.upper(), a.lower()), ctx)
            x, y = sorted((rnd.randint(0, la), rnd.randint(0, la)))
            check("substr_from/to/end", (A.substr_from(sk, U(x)).decrypt(ck), A.substr_to(sk, U(y)).decrypt(ck),
                                         A.substr_end(sk, U(x), U(y)).decrypt(ck)), (a[x:], a[:y], a[x:y]), ctx + (x, y))
            check("add", A.add(sk, E("!?")).decrypt(ck), a + "!?", ctx)
            m = rnd.randint(0, 3)
            check("repeat", E(p).repeat(sk, U(m), 3 * len(p) + 1).decrypt(ck), (p * m)[:3 * len(p) + 1], ctx + (m,))
            sv = A.split(sk, P)
            check("slices.len", sv.len(sk).decrypt(ck), len(parts), ctx)
        print(f"trial {t + 1}/{trials} {mode}: {bad} mismatch(es), {time.time() - t_start:.0f} s", flush=True)
sk.set_logdepth(False)
sk.close()
sys.exit(1 if bad else 0)
